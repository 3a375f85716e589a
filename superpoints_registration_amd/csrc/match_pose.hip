// a11 / a12 / a13 -- dual-softmax matching, Sinkhorn soft correspondences
// and the batched weighted Procrustes (Kabsch) solve on gfx950.
//
// Behaviour contract:
//   RegTR.softmax_correlation   /root/reference/src/models/qk_regtr_full.py:423-672
//     :453 correlation = F_s F_t^T / sqrt(D)
//     :457-459 / :565-567  attn = softmax(dim=-2) * softmax(dim=-1)
//     :468 / :576          val, ind = max over the longer side
//     :525-536 / :635-647  Sinkhorn affinity from the clamped score matrix
//   sinkhorn / compute_rigid_transform_with_sinkhorn  utils/se3_torch.py:166-239
//   compute_rigid_transform                           utils/se3_torch.py:109-163
//
// The reference loops over pairs in Python, materialises softmax copies of the
// N x M matrix and calls LAPACK through torch.svd (plus a device->host sync in
// an assert, se3_torch.py:132).  Here: one GEMM per pair (spr_linear's arithmetic:
// range-scaled split-fp16 MFMA by default, exact f32 in gemm mode 0) writes
// the score matrix once into scratch; row / column log-sum-exp passes stream
// it (wave per row, 64 columns x 4 row-lanes per workgroup for columns); the
// slack Sinkhorn is carried as two potential vectors
//     u_i = log(1 + sum_j exp(A_ij - v_j)),  v_j = log(1 + sum_i exp(A_ij - u_i))
// (algebraically identical to normalising the zero-padded (N+1)x(M+1) matrix,
// se3_torch.py:186-197), and the pose solve is one workgroup per pair with
// float64 moment accumulation and a 3x3 one-sided Jacobi SVD in registers --
// no host round trip anywhere.
#include <vector>

#include "attn_planes.h"
#include "spr_common.h"

namespace spr {
namespace {

struct PairDesc {
  int src_beg, n, tgt_beg, m;
  long long off;  // offset (floats) of this pair's N x M matrix in the scratch
};

// ---- row / column log-sum-exp over the scaled correlation ------------------
// lse over j of (c[i][j]*scale - sub[j])   (sub may be NULL); optionally +1
// inside the log-sum (the slack entry exp(0)).
__global__ void k_row_lse(const float* __restrict__ mat, const PairDesc* __restrict__ pd,
                          float* __restrict__ row_out /*packed by src token*/,
                          const float* __restrict__ col_sub /*packed by tgt token*/, int slack) {
  const PairDesc p = pd[blockIdx.y];
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= p.n) return;
  const float* r = mat + p.off + (size_t)row * p.m;
  float mx = slack ? 0.f : -INFINITY;
  for (int j = lane; j < p.m; j += 64) {
    const float v = r[j] - (col_sub ? col_sub[p.tgt_beg + j] : 0.f);
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < p.m; j += 64) {
    const float v = r[j] - (col_sub ? col_sub[p.tgt_beg + j] : 0.f);
    s += expf(v - mx);
  }
  s = wave_sum(s);
  if (slack) s += expf(0.f - mx);
  if (lane == 0) row_out[p.src_beg + row] = mx + logf(s);
}

// block = 1024 threads: 64 columns x 16 row lanes
constexpr int kColLanes = 16;
__global__ __launch_bounds__(1024) void k_col_lse(const float* __restrict__ mat,
                                                 const PairDesc* __restrict__ pd,
                                                 float* __restrict__ col_out,
                                                 const float* __restrict__ row_sub, int slack) {
  const PairDesc p = pd[blockIdx.y];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  __shared__ float smx[kColLanes][64], ssum[kColLanes][64];
  float mx = -INFINITY, s = 0.f;
  if (col < p.m) {
    for (int i = rl; i < p.n; i += kColLanes) {
      const float v = mat[p.off + (size_t)i * p.m + col] - (row_sub ? row_sub[p.src_beg + i] : 0.f);
      if (v > mx) {
        s = s * expf(mx - v) + 1.f;
        mx = v;
      } else {
        s += expf(v - mx);
      }
    }
  }
  smx[rl][cl] = mx;
  ssum[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && col < p.m) {
    float M = slack ? 0.f : -INFINITY;
    for (int k = 0; k < kColLanes; ++k) M = fmaxf(M, smx[k][cl]);
    float S = slack ? expf(0.f - M) : 0.f;
    for (int k = 0; k < kColLanes; ++k)
      if (smx[k][cl] > -INFINITY) S += ssum[k][cl] * expf(smx[k][cl] - M);
    col_out[p.tgt_beg + col] = M + logf(S);
  }
}

// ---- dual softmax arg-max ---------------------------------------------------
// N > M : for every tgt j   arg max_i  exp(c - col_lse[j]) * exp(c - row_lse[i])
// else  : for every src i   arg max_j  (same product)
__global__ __launch_bounds__(1024) void k_match_cols(const float* __restrict__ mat,
                                                    const PairDesc* __restrict__ pd,
                                                    const float* __restrict__ row_lse,
                                                    const float* __restrict__ col_lse,
                                                    float* __restrict__ val, int* __restrict__ ind) {
  const PairDesc p = pd[blockIdx.y];
  if (!(p.n > p.m)) return;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  __shared__ float sv[kColLanes][64];
  __shared__ int si[kColLanes][64];
  float best = -1.f;
  int bi = 0;
  if (col < p.m) {
    const float cl_j = col_lse[p.tgt_beg + col];
    for (int i = rl; i < p.n; i += kColLanes) {
      const float c = mat[p.off + (size_t)i * p.m + col];
      const float a = expf(c - cl_j) * expf(c - row_lse[p.src_beg + i]);
      if (a > best) {
        best = a;
        bi = i;
      }
    }
  }
  sv[rl][cl] = best;
  si[rl][cl] = bi;
  __syncthreads();
  if (rl == 0 && col < p.m) {
    for (int k = 1; k < kColLanes; ++k) {
      const float b = sv[k][cl];
      const int i = si[k][cl];
      if (b > best || (b == best && i < bi)) {
        best = b;
        bi = i;
      }
    }
    val[p.tgt_beg + col] = best;
    ind[p.tgt_beg + col] = bi;
  }
}

__global__ void k_match_rows(const float* __restrict__ mat, const PairDesc* __restrict__ pd,
                             const float* __restrict__ row_lse, const float* __restrict__ col_lse,
                             float* __restrict__ val, int* __restrict__ ind) {
  const PairDesc p = pd[blockIdx.y];
  if (p.n > p.m) return;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= p.n) return;
  const float rl_i = row_lse[p.src_beg + row];
  float best = -1.f;
  int bj = 0;
  for (int j = lane; j < p.m; j += 64) {
    const float c = mat[p.off + (size_t)row * p.m + j];
    const float a = expf(c - col_lse[p.tgt_beg + j]) * expf(c - rl_i);
    if (a > best) {
      best = a;
      bj = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float b = __shfl_xor(best, o, 64);
    const int j = __shfl_xor(bj, o, 64);
    if (b > best || (b == best && j < bj)) {
      best = b;
      bj = j;
    }
  }
  if (lane == 0) {
    val[p.src_beg + row] = best;
    ind[p.src_beg + row] = bj;
  }
}

// ---- elementwise transforms of the score matrix -----------------------------
__global__ void k_scale(float* __restrict__ mat, long long total, float s) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) mat[i] *= s;
}
// affinity = -(max(c*scale, 0) - sp_alpha) * inv_den      (qk_regtr_full.py:532-535)
// alpha, beta: the model's learnable scalars, read from DEVICE memory (no host round trip);
// softplus with threshold 20 like torch.nn.Softplus, evaluated in float64.
__global__ void k_affinity(float* __restrict__ mat, long long total, float scale,
                           const float* __restrict__ alpha_p, const float* __restrict__ beta_p) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const float alpha = alpha_p[0], beta = beta_p[0];
  const float sp_alpha = (float)(alpha > 20.f ? (double)alpha : log1p(exp((double)alpha)));
  const float inv_den = (float)(1.0 / (exp((double)beta) + 0.02));
  if (i < total) {
    const float sc = fmaxf(mat[i] * scale, 0.f);
    mat[i] = -(sc - sp_alpha) * inv_den;
  }
}

// w_i = sum_j P_ij ; t_hat_i = sum_j P_ij tgt_j / (w_i + 1e-6),  P = exp(A - u_i - v_j)
__global__ void k_sinkhorn_final(const float* __restrict__ mat, const PairDesc* __restrict__ pd,
                                 const float* __restrict__ u, const float* __restrict__ v,
                                 const float* __restrict__ xyz, float* __restrict__ out_w,
                                 float* __restrict__ out_t) {
  const PairDesc p = pd[blockIdx.y];
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= p.n) return;
  const float ui = u[p.src_beg + row];
  float w = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
  for (int j = lane; j < p.m; j += 64) {
    const float pij = expf(mat[p.off + (size_t)row * p.m + j] - ui - v[p.tgt_beg + j]);
    const float* t = xyz + 3 * (size_t)(p.tgt_beg + j);
    w += pij;
    tx += pij * t[0];
    ty += pij * t[1];
    tz += pij * t[2];
  }
  w = wave_sum(w);
  tx = wave_sum(tx);
  ty = wave_sum(ty);
  tz = wave_sum(tz);
  if (lane == 0) {
    // packed by src token; src tokens of pair b start at src_beg (global)
    const float d = w + 1e-6f;
    out_w[p.src_beg + row] = w;
    out_t[3 * (size_t)(p.src_beg + row) + 0] = tx / d;
    out_t[3 * (size_t)(p.src_beg + row) + 1] = ty / d;
    out_t[3 * (size_t)(p.src_beg + row) + 2] = tz / d;
  }
}

// ---- weighted Procrustes ------------------------------------------------------
__device__ void block_reduce_d(double* vals, int nvals, double* sh /*[256]*/) {
  // reduces each of vals[0..nvals) over the 256 threads; result in all threads
  for (int k = 0; k < nvals; ++k) {
    double x = wave_sum_d(vals[k]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    vals[k] = sh[0] + sh[1] + sh[2] + sh[3];
  }
}

__device__ void svd3_jacobi(const double A[3][3], double U[3][3], double S[3], double V[3][3]) {
  double G[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      G[i][j] = A[i][j];
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < 3; ++k) {
          al += G[k][p] * G[k][p];
          be += G[k][q] * G[k][q];
          ga += G[k][p] * G[k][q];
        }
        if (ga == 0.0 || fabs(ga) <= 1e-30 * sqrt(al * be)) continue;
        off = fmax(off, fabs(ga) / sqrt(al * be + 1e-300));
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; ++k) {
          const double gp = G[k][p], gq = G[k][q];
          G[k][p] = c * gp - s * gq;
          G[k][q] = s * gp + c * gq;
          const double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int j = 0; j < 3; ++j) S[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
  // sort descending (torch.svd order)
  int ord[3] = {0, 1, 2};
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (S[ord[b]] > S[ord[a]]) {
        int t = ord[a];
        ord[a] = ord[b];
        ord[b] = t;
      }
  double Gs[3][3], Vs[3][3], Ss[3];
  for (int j = 0; j < 3; ++j) {
    Ss[j] = S[ord[j]];
    for (int i = 0; i < 3; ++i) {
      Gs[i][j] = G[i][ord[j]];
      Vs[i][j] = V[i][ord[j]];
    }
  }
  const double tiny = 1e-14 * (Ss[0] > 0 ? Ss[0] : 1.0);
  for (int j = 0; j < 3; ++j) {
    S[j] = Ss[j];
    for (int i = 0; i < 3; ++i) {
      V[i][j] = Vs[i][j];
      U[i][j] = Ss[j] > tiny ? Gs[i][j] / Ss[j] : 0.0;
    }
  }
  // complete U for (numerically) rank deficient input
  if (!(S[0] > tiny)) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.0 : 0.0;
    return;
  }
  if (!(S[1] > tiny)) {
    // any unit vector orthogonal to u0
    int m = 0;
    if (fabs(U[1][0]) < fabs(U[m][0])) m = 1;
    if (fabs(U[2][0]) < fabs(U[m][0])) m = 2;
    double e[3] = {0, 0, 0};
    e[m] = 1.0;
    const double d = U[m][0];
    double n2 = 0;
    for (int i = 0; i < 3; ++i) {
      U[i][1] = e[i] - d * U[i][0];
      n2 += U[i][1] * U[i][1];
    }
    n2 = sqrt(n2);
    for (int i = 0; i < 3; ++i) U[i][1] /= n2;
  }
  if (!(S[2] > tiny)) {
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
}

__global__ __launch_bounds__(256) void k_procrustes(const float* __restrict__ a,
                                                    const float* __restrict__ b,
                                                    const float* __restrict__ w,
                                                    const int* __restrict__ pair_cu,
                                                    float* __restrict__ out) {
  const int pr = blockIdx.x;
  const int beg = pair_cu[pr], end = pair_cu[pr + 1];
  __shared__ double sh[256];
  double acc[7] = {0, 0, 0, 0, 0, 0, 0};  // sum w, sum w*a (3), sum w*b (3)
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    const double wi = w ? (double)w[i] : 1.0;
    acc[0] += wi;
    for (int d = 0; d < 3; ++d) {
      acc[1 + d] += wi * (double)a[3 * (size_t)i + d];
      acc[4 + d] += wi * (double)b[3 * (size_t)i + d];
    }
  }
  block_reduce_d(acc, 7, sh);
  // se3_torch.py:136-139: w~ = w / clamp_min(sum w, 1e-6); unweighted: mean
  double den = w ? fmax(acc[0], 1e-6) : fmax(acc[0], 1.0);
  double ca[3], cb[3];
  for (int d = 0; d < 3; ++d) {
    ca[d] = acc[1 + d] / den;
    cb[d] = acc[4 + d] / den;
  }
  double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    const double wi = (w ? (double)w[i] : 1.0) / den;
    double da[3], db[3];
    for (int d = 0; d < 3; ++d) {
      da[d] = (double)a[3 * (size_t)i + d] - ca[d];
      db[d] = ((double)b[3 * (size_t)i + d] - cb[d]) * wi;
    }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) cov[3 * r + c] += da[r] * db[c];
  }
  block_reduce_d(cov, 9, sh);
  if (threadIdx.x == 0) {
    double A[3][3], U[3][3], S[3], V[3][3];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) A[r][c] = cov[3 * r + c];
    svd3_jacobi(A, U, S, V);
    // R = V U^T, flip V[:,2] when det <= 0  (se3_torch.py:150-157)
    double R[3][3];
    for (int pass = 0; pass < 2; ++pass) {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
          R[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
      const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) -
                         R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                         R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
      if (det > 0.0) break;
      for (int r = 0; r < 3; ++r) V[r][2] = -V[r][2];
    }
    float* o = out + 12 * (size_t)pr;
    for (int r = 0; r < 3; ++r) {
      double t = cb[r];
      for (int c = 0; c < 3; ++c) {
        o[4 * r + c] = (float)R[r][c];
        t -= R[r][c] * ca[c];
      }
      o[4 * r + 3] = (float)t;
    }
  }
}

int build_pairs(const int* cu_host, int npairs, PairDesc* h, long long* total) {
  long long off = 0;
  for (int b = 0; b < npairs; ++b) {
    h[b].src_beg = cu_host[b];
    h[b].n = cu_host[b + 1] - cu_host[b];
    h[b].tgt_beg = cu_host[npairs + b];
    h[b].m = cu_host[npairs + b + 1] - cu_host[npairs + b];
    h[b].off = off;
    off += (long long)h[b].n * h[b].m;
    off = (off + 63) / 64 * 64;
  }
  *total = off;
  return 0;
}

size_t match_ws_bytes(const int* cu_host, int npairs) {
  long long off = 0;
  int tmax = cu_host[2 * npairs];
  for (int b = 0; b < npairs; ++b) {
    off += (long long)(cu_host[b + 1] - cu_host[b]) *
           (cu_host[npairs + b + 1] - cu_host[npairs + b]);
    off = (off + 63) / 64 * 64;
  }
  return align_up((size_t)off * 4, 256) + align_up(sizeof(PairDesc) * npairs, 256) +
         4 * align_up((size_t)tmax * 4, 256) + 2 * align_up(kAmaxParts * sizeof(float), 256) + 1024;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_match_workspace_bytes(const int* cu_host, int npairs) {
  return match_ws_bytes(cu_host, npairs);
}
extern "C" size_t spr_sinkhorn_workspace_bytes(const int* cu_host, int npairs) {
  return match_ws_bytes(cu_host, npairs);
}

namespace {
// Fills the scratch with raw correlations F_s F_t^T (unscaled), one MFMA GEMM
// per pair; uploads the pair descriptors.
__global__ void k_build_pairs(const int* __restrict__ cu, int npairs, PairDesc* pd) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long long off = 0;
  for (int b = 0; b < npairs; ++b) {
    PairDesc p;
    p.src_beg = cu[b];
    p.n = cu[b + 1] - cu[b];
    p.tgt_beg = cu[npairs + b];
    p.m = cu[npairs + b + 1] - cu[npairs + b];
    p.off = off;
    off += (long long)p.n * p.m;
    off = (off + 63) / 64 * 64;
    pd[b] = p;
  }
}

int correlate(const float* feat, int d, const int* cu_dev, const int* cu_host, int npairs, Workspace& w,
              float** mat_out, PairDesc** pd_out, std::vector<PairDesc>& h, long long* total,
              int* max_n, int* max_m, hipStream_t stream) {
  h.resize(npairs);
  build_pairs(cu_host, npairs, h.data(), total);
  float* mat = w.take<float>((size_t)*total);
  PairDesc* pd = w.take<PairDesc>(npairs);
  SPR_REQUIRE(mat && pd, "match: workspace carve failed");
  // descriptors are rebuilt on the device from cu (no pageable host copy)
  hipLaunchKernelGGL(k_build_pairs, dim3(1), dim3(64), 0, stream, cu_dev, npairs, pd);
  *max_n = 0;
  *max_m = 0;
  // two range measurements serve every pair: all src tokens (A operands), all tgt tokens (B)
  const float *sparts = nullptr, *tparts = nullptr;
  if (gemm_mode() == 1) {
    float* ps = w.take<float>(kAmaxParts);
    float* pt = w.take<float>(kAmaxParts);
    SPR_REQUIRE(pt != nullptr, "match: workspace carve failed");
    const int nsrc = cu_host[npairs], ntot = cu_host[2 * npairs];
    if (int rc = launch_absmax(feat, nsrc, d, d, ps, stream)) return rc;
    if (int rc = launch_absmax(feat + (size_t)nsrc * d, ntot - nsrc, d, d, pt, stream)) return rc;
    sparts = ps;
    tparts = pt;
  }
  for (int b = 0; b < npairs; ++b) {
    SPR_REQUIRE(h[b].n > 0 && h[b].m > 0, "match: empty cloud in pair %d", b);
    if (launch_linear_ranged(feat + (size_t)h[b].src_beg * d, h[b].n, d, feat + (size_t)h[b].tgt_beg * d,
                             h[b].m, nullptr, mat + h[b].off, sparts, tparts, stream))
      return 1;
    *max_n = h[b].n > *max_n ? h[b].n : *max_n;
    *max_m = h[b].m > *max_m ? h[b].m : *max_m;
  }
  *mat_out = mat;
  *pd_out = pd;
  return 0;
}
}  // namespace

extern "C" int spr_match_dualsoftmax(const float* feat, int d, const int* cu, const int* cu_host,
                                     int npairs, float* match_val, int* match_ind, void* ws,
                                     size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(npairs >= 1 && d % 32 == 0, "match: need npairs >= 1 and d %% 32 == 0");
  SPR_REQUIRE(ws_bytes >= match_ws_bytes(cu_host, npairs), "match: workspace too small");
  Workspace w(ws, ws_bytes);
  float* mat;
  PairDesc* pd;
  std::vector<PairDesc> h;
  long long total;
  int max_n, max_m;
  if (correlate(feat, d, cu, cu_host, npairs, w, &mat, &pd, h, &total, &max_n, &max_m, stream)) return 1;
  const int T = cu_host[2 * npairs];
  float* row_lse = w.take<float>(T);
  float* col_lse = w.take<float>(T);
  SPR_REQUIRE(col_lse != nullptr, "match: workspace carve failed");
  const float scale = 1.0f / sqrtf((float)d);
  hipLaunchKernelGGL(k_scale, dim3(cdiv(total, 256)), dim3(256), 0, stream, mat, total, scale);
  hipLaunchKernelGGL(k_row_lse, dim3(cdiv((long)max_n * 64, 256), npairs), dim3(256), 0, stream, mat,
                     pd, row_lse, (const float*)nullptr, 0);
  hipLaunchKernelGGL(k_col_lse, dim3(cdiv(max_m, 64), npairs), dim3(1024), 0, stream, mat, pd, col_lse,
                     (const float*)nullptr, 0);
  hipLaunchKernelGGL(k_match_cols, dim3(cdiv(max_m, 64), npairs), dim3(1024), 0, stream, mat, pd,
                     row_lse, col_lse, match_val, match_ind);
  hipLaunchKernelGGL(k_match_rows, dim3(cdiv((long)max_n * 64, 256), npairs), dim3(256), 0, stream,
                     mat, pd, row_lse, col_lse, match_val, match_ind);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_sinkhorn_correspondences(const float* feat, int d, const float* xyz, const int* cu,
                                            const int* cu_host, int npairs, const float* alpha,
                                            const float* beta, int n_iters, int slack, float* out_w,
                                            float* out_that,
                                            void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)slack;  // the reference's sinkhorn() always pads the slack row/col (se3_torch.py:182-184)
  SPR_REQUIRE(npairs >= 1 && d % 32 == 0 && n_iters >= 0, "sinkhorn: bad arguments");
  SPR_REQUIRE(ws_bytes >= match_ws_bytes(cu_host, npairs), "sinkhorn: workspace too small");
  Workspace w(ws, ws_bytes);
  float* mat;
  PairDesc* pd;
  std::vector<PairDesc> h;
  long long total;
  int max_n, max_m;
  if (correlate(feat, d, cu, cu_host, npairs, w, &mat, &pd, h, &total, &max_n, &max_m, stream)) return 1;
  const int T = cu_host[2 * npairs];
  float* u = w.take<float>(T);
  float* v = w.take<float>(T);
  SPR_REQUIRE(v != nullptr, "sinkhorn: workspace carve failed");
  const float scale = 1.0f / sqrtf((float)d);
  SPR_REQUIRE(alpha != nullptr && beta != nullptr, "sinkhorn: alpha / beta must be device pointers");
  hipLaunchKernelGGL(k_affinity, dim3(cdiv(total, 256)), dim3(256), 0, stream, mat, total, scale, alpha, beta);
  SPR_HIP_CHECK(hipMemsetAsync(u, 0, sizeof(float) * T, stream));
  SPR_HIP_CHECK(hipMemsetAsync(v, 0, sizeof(float) * T, stream));
  for (int it = 0; it < n_iters; ++it) {
    hipLaunchKernelGGL(k_row_lse, dim3(cdiv((long)max_n * 64, 256), npairs), dim3(256), 0, stream,
                       mat, pd, u, (const float*)v, 1);
    hipLaunchKernelGGL(k_col_lse, dim3(cdiv(max_m, 64), npairs), dim3(1024), 0, stream, mat, pd, v,
                       (const float*)u, 1);
  }
  hipLaunchKernelGGL(k_sinkhorn_final, dim3(cdiv((long)max_n * 64, 256), npairs), dim3(256), 0,
                     stream, mat, pd, u, v, xyz, out_w, out_that);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_weighted_procrustes(const float* a, const float* b, const float* w,
                                       const int* pair_cu, int npairs, float* out_pose,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(npairs >= 1, "procrustes: npairs must be >= 1");
  hipLaunchKernelGGL(k_procrustes, dim3(npairs), dim3(256), 0, stream, a, b, w, pair_cu, out_pose);
  SPR_LAUNCH_CHECK();
  return 0;
}
