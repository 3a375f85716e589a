// Loss terms of RegTR.compute_loss (forward) on gfx950 -- SURVEY section 8f row 1.
//
// Behaviour contract (reference, /root/reference/src):
//   compute_overlaps            models/backbone_kpconv/kpconv.py:552-578
//       ground-truth overlap averaged up the pooling pyramid, clamp to [0, 1]
//   nn.BCEWithLogitsLoss        models/qk_regtr_full.py:90, :329 (mean reduction)
//   InfoNCELossFull             models/losses/feature_loss.py:246-314
//       logits = A triu-symmetrised(W) B^T; positive = nearest target keypoint if
//       closer than r_p; targets closer than r_n (except the positive) ignored;
//       loss = mean over anchors with a positive of (logsumexp - positive logit)
//   transform loss              qk_regtr_full.py:349-355
//       sum over pairs of mean |T_gt x - T_pred x| over the pair's source keypoints
//
// All reductions are deterministic (fixed partition, float64 accumulation, no
// atomics).  Distances use direct differences; torch.cdist's matmul expansion
// differs from that by rounding only (a keypoint pair within ~1e-6 of a radius
// could be classified differently).
#include <vector>

#include "spr_common.h"

namespace spr {
namespace {

constexpr int RB = 256;   // reduction block

// out[q] = clamp(sum_valid ov[idx] / count_valid, 0, 1); 0/0 -> NaN like the reference
__global__ void k_overlap_pool(const float* __restrict__ ov, int ns, const int* __restrict__ pool,
                               int stride, int w, int nq, float* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  float s = 0.f;
  int c = 0;
  for (int k = 0; k < w; ++k) {
    const int id = pool[(size_t)q * stride + k];
    if (id >= 0 && id < ns) {
      s += ov[id];
      ++c;
    }
  }
  const float v = s / (float)c;
  out[q] = v != v ? v : fminf(fmaxf(v, 0.f), 1.f);
}

__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  __syncthreads();
  return t;   // valid in thread 0
}

// partial[b] = sum over the block's slice of  max(x,0) - x y + log1p(exp(-|x|))
__global__ void k_bce_partial(const float* __restrict__ x, const float* __restrict__ y, int n,
                              double* __restrict__ partial) {
  __shared__ double sh[RB / 64];
  const int i = blockIdx.x * RB + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    const float xi = x[i], yi = y[i];
    v = (double)(fmaxf(xi, 0.f) - xi * yi + log1pf(expf(-fabsf(xi))));
  }
  const double t = block_sum(v, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// out[0] = scale * sum(partial[0..n)) / (den ? sum(den_partial) : 1)
__global__ void k_finish(const double* __restrict__ partial, const double* __restrict__ den, int n,
                         double scale, float* __restrict__ out) {
  __shared__ double sh[RB / 64];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += RB) {
    a += partial[i];
    if (den) b += den[i];
  }
  const double ta = block_sum(a, sh);
  const double tb = block_sum(b, sh);
  if (threadIdx.x == 0) out[0] = (float)(den ? ta / tb : ta * scale);
}

// W_sym = triu(W) + triu(W)^T
__global__ void k_wsym(const float* __restrict__ W, int d, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d * d) return;
  const int r = i / d, c = i % d;
  const float a = c >= r ? W[r * d + c] : 0.f;   // triu(W)[r][c]
  const float b = r >= c ? W[c * d + r] : 0.f;   // triu(W)^T[r][c]
  out[i] = a + b;
}

// anchors transformed by the ground-truth pose (se3_torch.py:16-35)
__global__ void k_transform(const float* __restrict__ pose, const float* __restrict__ xyz, int n,
                            float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    out[3 * i + r] = (x * pose[4 * r] + y * pose[4 * r + 1] + z * pose[4 * r + 2]) + pose[4 * r + 3];
}

// One wave per anchor row: nearest positive, ignore set, logsumexp, positive logit.
// row_loss[i] = mask ? lse - logit[i, idx1] : 0 ; row_mask[i] = mask
__global__ void k_infonce_rows(const float* __restrict__ logits, int n, int m,
                               const float* __restrict__ a_xyz, const float* __restrict__ p_xyz,
                               float r_p, float r_n, float* __restrict__ row_loss,
                               float* __restrict__ row_mask) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float ax = a_xyz[3 * i], ay = a_xyz[3 * i + 1], az = a_xyz[3 * i + 2];
  const float* row = logits + (size_t)i * m;
  // pass 1: nearest positive (lowest index on exact ties)
  float best = INFINITY;
  int bj = 0x7fffffff;
  for (int j = lane; j < m; j += 64) {
    const float dx = ax - p_xyz[3 * j], dy = ay - p_xyz[3 * j + 1], dz = az - p_xyz[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (d < best || (d == best && j < bj)) {
      best = d;
      bj = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (ob < best || (ob == best && oj < bj)) {
      best = ob;
      bj = oj;
    }
  }
  // pass 2: logsumexp over the targets that are not ignored
  float mx = -INFINITY;
  for (int j = lane; j < m; j += 64) {
    const float dx = ax - p_xyz[3 * j], dy = ay - p_xyz[3 * j + 1], dz = az - p_xyz[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(d < r_n) || j == bj) mx = fmaxf(mx, row[j]);
  }
  mx = wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < m; j += 64) {
    const float dx = ax - p_xyz[3 * j], dy = ay - p_xyz[3 * j + 1], dz = az - p_xyz[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(d < r_n) || j == bj) se += expf(row[j] - mx);
  }
  se = wave_sum(se);
  if (lane == 0) {
    const bool mask = best < r_p;
    row_loss[i] = mask ? (mx + logf(se)) - row[bj] : 0.f;
    row_mask[i] = mask ? 1.f : 0.f;
  }
}

// partial sums of two float arrays (loss, mask) in double
__global__ void k_pair_partial(const float* __restrict__ a, const float* __restrict__ b, int n,
                               double* __restrict__ pa, double* __restrict__ pb) {
  __shared__ double sh[RB / 64];
  const int i = blockIdx.x * RB + threadIdx.x;
  const double va = i < n ? (double)a[i] : 0.0, vb = i < n ? (double)b[i] : 0.0;
  const double ta = block_sum(va, sh);
  const double tb = block_sum(vb, sh);
  if (threadIdx.x == 0) {
    pa[blockIdx.x] = ta;
    pb[blockIdx.x] = tb;
  }
}

// partial[b] = sum over slice of |T_gt x - T_pred x| (all three components)
__global__ void k_tloss_partial(const float* __restrict__ pose_gt, const float* __restrict__ pose_pred,
                                const float* __restrict__ xyz, int n, double* __restrict__ partial) {
  __shared__ double sh[RB / 64];
  const int i = blockIdx.x * RB + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float g = (x * pose_gt[4 * r] + y * pose_gt[4 * r + 1] + z * pose_gt[4 * r + 2]) + pose_gt[4 * r + 3];
      const float p = (x * pose_pred[4 * r] + y * pose_pred[4 * r + 1] + z * pose_pred[4 * r + 2]) + pose_pred[4 * r + 3];
      v += (double)fabsf(g - p);
    }
  }
  const double t = block_sum(v, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// ---- backward kernels ----------------------------------------------------------
// d/dx of mean(BCE-with-logits) = (sigmoid(x) - y) / n, times the upstream scalar gradient gout[0]
__global__ void k_bce_bwd(const float* __restrict__ x, const float* __restrict__ y, int n,
                          const float* __restrict__ gout, float* __restrict__ dx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = 1.f / (1.f + expf(-x[i]));
  dx[i] = gout[0] * (s - y[i]) / (float)n;
}

// InfoNCE (feature_loss.py:268-296): row i with a positive contributes lse_i - logit[i, pos_i];
// d logits[i, j] = mask_i (softmax over the non-ignored targets - [j == pos_i]).  Un-normalised:
// the caller divides by the number of masked rows (row_mask is written too).
__global__ void k_infonce_dlogits(const float* __restrict__ logits, int n, int m,
                                  const float* __restrict__ a_xyz, const float* __restrict__ p_xyz,
                                  float r_p, float r_n, float* __restrict__ dlogits,
                                  float* __restrict__ row_mask) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  const float ax = a_xyz[3 * i], ay = a_xyz[3 * i + 1], az = a_xyz[3 * i + 2];
  const float* row = logits + (size_t)i * m;
  float* drow = dlogits + (size_t)i * m;
  float best = INFINITY;
  int bj = 0x7fffffff;
  for (int j = lane; j < m; j += 64) {
    const float dx = ax - p_xyz[3 * j], dy = ay - p_xyz[3 * j + 1], dz = az - p_xyz[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (d < best || (d == best && j < bj)) {
      best = d;
      bj = j;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oj = __shfl_xor(bj, o, 64);
    if (ob < best || (ob == best && oj < bj)) {
      best = ob;
      bj = oj;
    }
  }
  const bool mask = best < r_p;
  float mx = -INFINITY;
  for (int j = lane; j < m; j += 64) {
    const float dx = ax - p_xyz[3 * j], dy = ay - p_xyz[3 * j + 1], dz = az - p_xyz[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(d < r_n) || j == bj) mx = fmaxf(mx, row[j]);
  }
  mx = wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < m; j += 64) {
    const float dx = ax - p_xyz[3 * j], dy = ay - p_xyz[3 * j + 1], dz = az - p_xyz[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(d < r_n) || j == bj) se += expf(row[j] - mx);
  }
  se = wave_sum(se);
  const float inv = 1.f / se;
  for (int j = lane; j < m; j += 64) {
    const float dx = ax - p_xyz[3 * j], dy = ay - p_xyz[3 * j + 1], dz = az - p_xyz[3 * j + 2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    float g = 0.f;
    if (mask && (!(d < r_n) || j == bj)) g = expf(row[j] - mx) * inv;
    if (mask && j == bj) g -= 1.f;
    drow[j] = g;
  }
  if (lane == 0) row_mask[i] = mask ? 1.f : 0.f;
}

// dW[r][c] = dWsym[r][c] + dWsym[c][r] for c >= r, 0 below the diagonal  (W_sym = triu(W) + triu(W)^T)
__global__ void k_wsym_bwd(const float* __restrict__ dws, int d, float* __restrict__ dW) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d * d) return;
  const int r = i / d, c = i % d;
  dW[i] = c >= r ? dws[r * d + c] + dws[c * d + r] : 0.f;
}

// transform loss of one pair: mean over points and components of |T_gt x - T_pred x|;
// d/d pose_pred[r][c] = -(gout/(3n)) sum_i sign(g - p)_r x_c   (c = 3: x_c = 1).  One workgroup.
__global__ __launch_bounds__(RB) void k_tloss_bwd(const float* __restrict__ pose_gt,
                                                  const float* __restrict__ pose_pred,
                                                  const float* __restrict__ xyz, int n,
                                                  const float* __restrict__ gout, float* __restrict__ dpose) {
  __shared__ double sh[RB / 64];
  double acc[12];
  for (int k = 0; k < 12; ++k) acc[k] = 0.0;
  for (int i = threadIdx.x; i < n; i += RB) {
    const float x[4] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 1.f};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float g = (x[0] * pose_gt[4 * r] + x[1] * pose_gt[4 * r + 1] + x[2] * pose_gt[4 * r + 2]) + pose_gt[4 * r + 3];
      const float p = (x[0] * pose_pred[4 * r] + x[1] * pose_pred[4 * r + 1] + x[2] * pose_pred[4 * r + 2]) + pose_pred[4 * r + 3];
      const float df = g - p;
      const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * r + c] -= (double)(sgn * x[c]);
    }
  }
  for (int k = 0; k < 12; ++k) {
    const double t = block_sum(acc[k], sh);
    if (threadIdx.x == 0) dpose[k] = (float)(t * (double)gout[0] / (3.0 * (double)n));
  }
}

// out[0] = sum_b pair[b]
__global__ void k_sum_small(const float* __restrict__ pair, int n, float scale, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int b = 0; b < n; ++b) s += (double)pair[b];
  out[0] = (float)(s * scale);
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" int spr_overlap_pool(const float* ov_prev, int ns_prev, const int* pool, int pool_stride, int w,
                                int nq, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(nq >= 1 && w >= 1 && pool_stride >= w && ns_prev >= 1, "overlap_pool: bad sizes");
  hipLaunchKernelGGL(k_overlap_pool, dim3(cdiv(nq, 256)), dim3(256), 0, stream, ov_prev, ns_prev, pool,
                     pool_stride, w, nq, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spr_loss_workspace_bytes(int n_max, int m_max, int d) {
  if (n_max < 0 || m_max < 0 || d < 0) return 0;
  const size_t n = (size_t)(n_max > 0 ? n_max : 1), m = (size_t)(m_max > 0 ? m_max : 1);
  return align_up(n * m * 4, 256) + align_up(n * d * 4, 256) + align_up((size_t)d * d * 4, 256) +
         align_up(n * 12, 256) + 2 * align_up(n * 4, 256) + 2 * align_up((size_t)cdiv(n, RB) * 8, 256) +
         2 * spr_linear_workspace_bytes() + 1024;
}

extern "C" int spr_bce_logits_mean(const float* x, const float* y, int n, float* out, void* ws,
                                   size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1, "bce: n must be >= 1");
  const int nb = cdiv(n, RB);
  SPR_REQUIRE(ws != nullptr && ws_bytes >= (size_t)nb * 8, "bce: workspace too small");
  double* partial = (double*)ws;
  hipLaunchKernelGGL(k_bce_partial, dim3(nb), dim3(RB), 0, stream, x, y, n, partial);
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(RB), 0, stream, partial, (const double*)nullptr, nb, 1.0 / n, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_infonce_pair(const float* anchor_feat, int n, const float* positive_feat, int m, int d,
                                const float* anchor_xyz, const float* pose_gt, const float* positive_xyz,
                                const float* W, float r_p, float r_n, float* out, void* ws, size_t ws_bytes,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1 && m >= 1 && d >= 32 && d % 32 == 0, "infonce: bad sizes n=%d m=%d d=%d", n, m, d);
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_loss_workspace_bytes(n, m, d), "infonce: workspace too small");
  Workspace w(ws, ws_bytes);
  float* logits = w.take<float>((size_t)n * m);
  float* t = w.take<float>((size_t)n * d);
  float* wsym = w.take<float>((size_t)d * d);
  float* axyz = w.take<float>((size_t)n * 3);
  float* rl = w.take<float>(n);
  float* rm = w.take<float>(n);
  const int nb = cdiv(n, RB);
  double* pa = w.take<double>(nb);
  double* pb = w.take<double>(nb);
  const size_t lws = spr_linear_workspace_bytes();
  char* lw1 = w.take<char>(lws);
  char* lw2 = w.take<char>(lws);
  SPR_REQUIRE(pb != nullptr && lw2 != nullptr, "infonce: workspace carve failed");
  hipLaunchKernelGGL(k_wsym, dim3(cdiv(d * d, 256)), dim3(256), 0, stream, W, d, wsym);
  hipLaunchKernelGGL(k_transform, dim3(cdiv(n, 256)), dim3(256), 0, stream, pose_gt, anchor_xyz, n, axyz);
  // logits = (A W_sym) B^T : W_sym is symmetric, so the NT GEMM applies it as is
  if (int rc = spr_linear(anchor_feat, n, d, wsym, d, nullptr, nullptr, SPR_ACT_NONE, t, lw1, lws, stream_)) return rc;
  if (int rc = spr_linear(t, n, d, positive_feat, m, nullptr, nullptr, SPR_ACT_NONE, logits, lw2, lws, stream_)) return rc;
  hipLaunchKernelGGL(k_infonce_rows, dim3(cdiv((long)n * 64, 256)), dim3(256), 0, stream, logits, n, m, axyz,
                     positive_xyz, r_p, r_n, rl, rm);
  hipLaunchKernelGGL(k_pair_partial, dim3(nb), dim3(RB), 0, stream, rl, rm, n, pa, pb);
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(RB), 0, stream, pa, pb, nb, 1.0, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_transform_l1_pair(const float* pose_gt, const float* pose_pred, const float* xyz, int n,
                                     float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1, "transform_l1: n must be >= 1");
  const int nb = cdiv(n, RB);
  SPR_REQUIRE(ws != nullptr && ws_bytes >= (size_t)nb * 8, "transform_l1: workspace too small");
  double* partial = (double*)ws;
  hipLaunchKernelGGL(k_tloss_partial, dim3(nb), dim3(RB), 0, stream, pose_gt, pose_pred, xyz, n, partial);
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(RB), 0, stream, partial, (const double*)nullptr, nb, 1.0 / (3.0 * n),
                     out);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_sum_scaled(const float* values, int n, float scale, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1, "sum_scaled: n must be >= 1");
  hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(64), 0, stream, values, n, scale, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

// ---- backward entry points ---------------------------------------------------------------------
extern "C" int spr_bce_logits_mean_bwd(const float* x, const float* y, int n, const float* gout, float* dx,
                                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1 && x && y && gout && dx, "bce_bwd: bad arguments");
  hipLaunchKernelGGL(k_bce_bwd, dim3(cdiv(n, 256)), dim3(256), 0, stream, x, y, n, gout, dx);
  SPR_LAUNCH_CHECK();
  return 0;
}

// Step 1 of the InfoNCE backward of one pair: logits (recomputed like the forward) ->
// un-normalised d logits [n, m] and the row mask [n].  wsym_out [d, d] and t_out [n, d] (= A W_sym) are
// returned for the caller's GEMM chain (dA = (dlogits B) W_sym, dB = dlogits^T t, dW_sym = A^T (dlogits B)).
extern "C" int spr_infonce_pair_dlogits(const float* anchor_feat, int n, const float* positive_feat, int m, int d,
                                        const float* anchor_xyz, const float* pose_gt, const float* positive_xyz,
                                        const float* W, float r_p, float r_n, float* dlogits, float* row_mask,
                                        float* wsym_out, float* t_out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1 && m >= 1 && d >= 32 && d % 32 == 0, "infonce_bwd: bad sizes n=%d m=%d d=%d", n, m, d);
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_loss_workspace_bytes(n, m, d), "infonce_bwd: workspace too small");
  Workspace w(ws, ws_bytes);
  float* logits = w.take<float>((size_t)n * m);
  (void)w.take<float>((size_t)n * d);
  (void)w.take<float>((size_t)d * d);
  float* axyz = w.take<float>((size_t)n * 3);
  (void)w.take<float>(n);
  (void)w.take<float>(n);
  const int nb = cdiv(n, RB);
  (void)w.take<double>(nb);
  (void)w.take<double>(nb);
  const size_t lws = spr_linear_workspace_bytes();
  char* lw1 = w.take<char>(lws);
  char* lw2 = w.take<char>(lws);
  SPR_REQUIRE(lw2 != nullptr, "infonce_bwd: workspace carve failed");
  hipLaunchKernelGGL(k_wsym, dim3(cdiv(d * d, 256)), dim3(256), 0, stream, W, d, wsym_out);
  hipLaunchKernelGGL(k_transform, dim3(cdiv(n, 256)), dim3(256), 0, stream, pose_gt, anchor_xyz, n, axyz);
  if (int rc = spr_linear(anchor_feat, n, d, wsym_out, d, nullptr, nullptr, SPR_ACT_NONE, t_out, lw1, lws, stream_)) return rc;
  if (int rc = spr_linear(t_out, n, d, positive_feat, m, nullptr, nullptr, SPR_ACT_NONE, logits, lw2, lws, stream_)) return rc;
  hipLaunchKernelGGL(k_infonce_dlogits, dim3(cdiv((long)n * 64, 256)), dim3(256), 0, stream, logits, n, m, axyz,
                     positive_xyz, r_p, r_n, dlogits, row_mask);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_wsym_bwd(const float* dwsym, int d, float* dW, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(d >= 1 && dwsym && dW, "wsym_bwd: bad arguments");
  hipLaunchKernelGGL(k_wsym_bwd, dim3(cdiv(d * d, 256)), dim3(256), 0, stream, dwsym, d, dW);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_transform_l1_pair_bwd(const float* pose_gt, const float* pose_pred, const float* xyz, int n,
                                         const float* gout, float* dpose_pred, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1 && pose_gt && pose_pred && xyz && gout && dpose_pred, "transform_l1_bwd: bad arguments");
  hipLaunchKernelGGL(k_tloss_bwd, dim3(1), dim3(RB), 0, stream, pose_gt, pose_pred, xyz, n, gout, dpose_pred);
  SPR_LAUNCH_CHECK();
  return 0;
}
