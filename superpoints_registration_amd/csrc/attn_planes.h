// Layout of the split-fp16 operand planes the attention kernel consumes
// (attention.hip) and the in-projection GEMM can emit directly (linear.hip).
#pragma once
#include "spr_common.h"

namespace spr {

// Token columns of segment s in the transposed V planes start at a 16-byte
// aligned column; gaps and the tail hold zeros.
__host__ __device__ inline int attn_vstart_of(int cu_s, int s) { return (cu_s + 8 * s) & ~7; }

struct AttnPlanes {
  _Float16 *qh, *ql, *kh, *kl;   // [nhead][T][32] head-major, Q pre-scaled by log2(e)/sqrt(d)
  _Float16 *vth, *vtl;           // [nhead*32][tp] transposed
  const int* cu;                 // [nseg + 1]
  int nseg, t_total, tp;
  float qscale;
};

// Split-fp16 GEMM  planes <- x [m, k] . w [n, k]^T + bias  for the n output
// features [f0, f0 + n) of the packed in-projection (0..255 = Q, 256..511 = K,
// 512..767 = V; d_model = 256, head_dim = 32).  n and f0 are multiples of 256.
int launch_inproj_planes(const float* x, int m, int k, const float* w, int n, const float* bias, int f0,
                         const AttnPlanes& planes, hipStream_t stream);

// Exact-f32 / generic GEMM used by the mode-0 fallback of the fused entry point.
int launch_linear_plain(const float* x, int m, int k, const float* w, int n, const float* bias, float* out,
                        hipStream_t stream);

}  // namespace spr
