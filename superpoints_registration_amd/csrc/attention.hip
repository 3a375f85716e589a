// a9 -- varlen multi-head attention core (flash style, exact f32) on gfx950.
//
// Behaviour contract: the scaled-dot-product core of nn.MultiheadAttention as
// called four times per layer in TransformerCrossEncoderLayer.forward_pre
//   /root/reference/src/models/transformer/transformers.py:198-227
// (softmax(Q K^T / sqrt(head_dim)) V per head, key padding mask = padded
// positions excluded, dropout 0).  The reference pads every sequence to the
// batch maximum and materialises (B*heads, Lq, Lk) weights; here tokens stay
// packed, segment s reads the keys/values of segment kv_seg[s] (itself for
// self attention, the partner cloud for cross attention) and nothing of size
// Lq x Lk ever reaches memory.
//
// Kernel shape (head_dim = 32):
//   workgroup = 4 waves = 128 queries of one (segment, head); each wave owns
//   32 queries.  S^T = K Q^T is computed with v_mfma_f32_32x32x2_f32 so that
//   a query is a LANE: the softmax row reductions are in-register (16
//   accumulator registers + one cross-half shuffle), and the probabilities in
//   the accumulator are already the B operand of O^T = V^T P^T with the key
//   order permuted consistently on both operands -- no LDS round trip for P.
//   K / V tiles of 32 keys are staged in LDS (row stride 33 / 32 words ->
//   conflict-free fragment reads) and shared by the four waves.
#include "attn_planes.h"
#include "spr_common.h"
#include <atomic>
#include <type_traits>

namespace spr {
namespace {

constexpr int HD = 32;        // head dim
constexpr int KT = 32;        // keys per tile
constexpr int QW = 32;        // queries per wave
constexpr int QB = 128;       // queries per workgroup
constexpr int KS = HD + 1;    // K tile row stride (words)

__global__ __launch_bounds__(256) void k_attn(
    const float* __restrict__ q, int q_stride, const float* __restrict__ k, int k_stride,
    const float* __restrict__ v, int v_stride, const int* __restrict__ cu,
    const int* __restrict__ kv_seg, int nseg, int nhead, float scale, float* __restrict__ out,
    int o_stride) {
  // two LDS stages: tile t+1 is written while tile t is being read; one barrier per tile
  __shared__ float Ks[2][KT * KS];
  __shared__ float Vs[2][KT * HD];
  // 1-D grid: all query tiles of one (segment, head) -- which stream the same
  // K/V -- are placed on one XCD (ids b and b+8 share an L2)
  int seg, head, qt;
  {
    const int nqt = gridDim.x / (nhead * nseg);
    const int ngrp = nhead * nseg;
    const int b = blockIdx.x;
    if ((ngrp & 7) == 0) {
      const int xcd = b & 7, idx = b >> 3;
      const int g = xcd + 8 * (idx / nqt);
      qt = idx % nqt;
      head = g % nhead;
      seg = g / nhead;
    } else {
      qt = b % nqt;
      head = (b / nqt) % nhead;
      seg = b / (nqt * nhead);
    }
  }
  const int qbeg = cu[seg], qlen = cu[seg + 1] - qbeg;
  const int q0 = qt * QB;
  if (q0 >= qlen) return;  // uniform for the workgroup
  const int ks = kv_seg[seg];
  const int kbeg = cu[ks], klen = cu[ks + 1] - kbeg;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int qi = q0 + wave * QW + l31;  // this lane's query (local index)
  const bool qok = qi < qlen;
  const int hoff = head * HD;

  // Q^T as B operand of S^T = K Q^T:  B[k = d][col = query]; step s covers
  // d = 2s + lh.  Pre-scaled by log2(e)/sqrt(head_dim): softmax runs in base 2
  // (v_exp_f32 is a base-2 exponential), mathematically identical.
  float qreg[16];
  {
    const float sc = scale * 1.4426950408889634f;
    const float* qp = q + (size_t)(qbeg + (qok ? qi : 0)) * q_stride + hoff;
#pragma unroll
    for (int s = 0; s < 16; ++s) qreg[s] = qok ? qp[2 * s + lh] * sc : 0.f;
  }

  f32x16 o;  // O^T[d][query]: lane = query, d = (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging role of this thread: key row sr (0..31), 4 dims starting at sc4
  const int sr = tid >> 3, sc4 = (tid & 7) * 4;
  f32x4 kreg, vreg;
  // The prefetch must stay in flight across the MFMA block; hipcc sinks plain
  // loads to their first use, so the two loads are issued by inline asm and
  // retired by an explicit s_waitcnt (nothing else is on the vector-memory
  // queue inside the loop).
  auto fetch = [&](int kt) {
    const int r = min(kt + sr, klen - 1);          // clamp: always a valid row
    const size_t row = (size_t)(kbeg + r);
    const float* kp_ = k + row * k_stride + hoff + sc4;
    const float* vp_ = v + row * v_stride + hoff + sc4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kreg) : "v"(kp_));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vreg) : "v"(vp_));
  };
  auto stash = [&](int kt, int buf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const bool in = kt + sr < klen;                // rows past the segment read as zero
    float* kd = Ks[buf] + sr * KS + sc4;
    kd[0] = in ? kreg[0] : 0.f;
    kd[1] = in ? kreg[1] : 0.f;
    kd[2] = in ? kreg[2] : 0.f;
    kd[3] = in ? kreg[3] : 0.f;
    f32x4 vz = {in ? vreg[0] : 0.f, in ? vreg[1] : 0.f, in ? vreg[2] : 0.f, in ? vreg[3] : 0.f};
    *reinterpret_cast<f32x4*>(Vs[buf] + sr * HD + sc4) = vz;
  };

  if (klen > 0) {
    fetch(0);
    stash(0, 0);
  }
  __syncthreads();

  int buf = 0;
  for (int kt = 0; kt < klen; kt += KT, buf ^= 1) {
    const bool more = kt + KT < klen;
    if (more) fetch(kt + KT);
    __builtin_amdgcn_sched_barrier(0);

    // S^T tile: rows = keys, cols = queries.  A[row = key l31][k = d = 2s+lh]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    // all 16 A fragments are read from LDS up front (one exposed LDS latency
    // per tile instead of one per MFMA pair); V fragments are fetched before
    // the softmax so their latency hides under the VALU work.
    const float* kp = Ks[buf] + l31 * KS + lh;
    const float* vp = Vs[buf] + l31;
    float ka[16], va[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) ka[s] = kp[2 * s];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 16; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s], qreg[s], st, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) va[r] = vp[((r & 3) + 8 * (r >> 2) + 4 * lh) * HD];
    __builtin_amdgcn_sched_barrier(0);

    // lane holds, for its query, the 16 keys  j(r) = (r&3) + 8*(r>>2) + 4*lh
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (j >= klen) st[r] = -INFINITY;  // tail of the key segment
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // other half-lane: other 16 keys
    const float m_new = fmaxf(m_run, mx);    // finite: every tile has >= 1 key
    const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= corr;

    // O^T += V^T P^T.  Step r: lane half h contributes key j0(r) + 4h with
    // j0(r) = (r&3) + 8*(r>>2); B operand = st[r] (already in place),
    // A[row = d = l31][k = h] = V[j0(r) + 4h][d].
#pragma unroll
    for (int r = 0; r < 16; ++r)
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(va[r], st[r], o, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) stash(kt + KT, buf ^ 1);
    __syncthreads();
  }

  if (qok) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* op = out + (size_t)(qbeg + qi) * o_stride + hoff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      // registers 4g..4g+3 -> d = 8g + 4*lh + (0..3): one float4
      float4 w4 = make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv,
                              o[4 * g + 3] * inv);
      *reinterpret_cast<float4*>(op + 8 * g + 4 * lh) = w4;
    }
  }
}

// ---------------------------------------------------------------------------
// Split-fp16 variant ("h3"): fp32-level accuracy, ~5x less matrix-core time.
// Every operand x is carried as hi = fp16(x m) and lo = fp16(x m - hi) with a
// per-tensor multiplier m (below); a.b ~= ah.bh + ah.bl + al.bh in ONE fp32
// accumulator (the matrix cores honour fp16 subnormals, scripts/abl/denorm.hip).
//
// Range safety (k_plane_scales): K is multiplied by 2^ek and Q by
// log2(e)/sqrt(d) 2^-ek with ek = half the exponent gap between the bounds of
// |q| and |k| -- the scores are unchanged, both planes sit at sqrt(|q||k|) and
// cannot overflow unless the scores themselves exceed 2^30; V is multiplied by
// 2^ev (max |v| 2^ev in [2^14, 2^15)) and the output by 2^-ev.  The bounds are
// measured (unfused entry: launch_absmax on q, k, v) or derived (fused entry:
// max|x| * max row L1 norm of the projection block + max|bias|).
// Mode 2 ("h1") runs the same kernel with the hi planes only: single-pass fp16
// operands (11 significand bits), fp32 softmax and accumulators, 1/3 of the MFMAs.
//
// Two kernels:
//   k_attn_pack  splits Q (pre-scaled by log2(e)/sqrt(d)), K and V ONCE per
//     call into fp16 hi/lo planes in the workspace.  V is written TRANSPOSED
//     ([feature][token column]) because the V^T A-fragment of O^T = V^T P^T
//     wants 4 consecutive keys of one feature.  Token columns of segment s
//     start at vstart(s) = (cu[s] + 8 s) & ~7 (16-byte aligned rows for the
//     wide loads); gap and tail columns are written as zeros.
//   k_attn_h3    streams 64-key tiles of those planes through LDS (pure 16-byte
//     copies, no conversion in the loop) and runs, per wave of 32 queries:
//       S^T = K Q^T : 2 key sub-tiles x 2 k-steps x 3 v_mfma_f32_32x32x16_f16
//       online softmax, lane = query, base 2, lazy rescale; P is scaled by
//         2^10 so small probabilities stay in fp16's normal range (cancels in
//         the final 1/l), hi by packed RTZ conversion, lo by v_fma_mix*_f16
//       O^T = V^T P^T: the probabilities in the S^T accumulator are converted
//         in place -- registers 8s..8s+7 of a lane are exactly the B fragment
//         of k-step s with key order kappa(s,h,j) = 16s + 8(j>>2) + 4h + (j&3);
//         the V^T A-fragment is read with the same order (two 8-byte reads).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int KT2 = 64;  // keys per tile (two 32-key MFMA sub-tiles)
constexpr int KH = 40;   // K tile row stride in halves (80 B: conflict-free ds_read_b128)
constexpr int VH = 72;   // V^T tile row stride in halves (144 B: 2-pass ds_read_b64)
constexpr int PT = 64;   // token columns per pack workgroup
constexpr int PS = 72;   // pack transposition row stride (halves)

__device__ __forceinline__ int vstart(const int* __restrict__ cu, int s) {
  return attn_vstart_of(cu[s], s);
}

// Zeroes the gap columns between segments and the tail of the transposed V planes
// (written by k_attn_pack itself on the unfused path; needed when the in-projection
// GEMM emits the planes).  One workgroup per plane row.
__global__ __launch_bounds__(256) void k_attn_zero_gaps(const int* __restrict__ cu, int nseg, int tp,
                                                         _Float16* __restrict__ vth,
                                                         _Float16* __restrict__ vtl) {
  const int f = blockIdx.x, d = gridDim.x;
  for (int s = 0; s < nseg; ++s) {
    const int beg = vstart(cu, s) + cu[s + 1] - cu[s];
    const int end = s + 1 < nseg ? vstart(cu, s + 1) : tp;
    for (int c = beg + threadIdx.x; c < end; c += 256) {
      vth[attn_v_off(f, attn_vperm(c), d)] = (_Float16)0.f;     // (a bijection inside every aligned 16-column group)
      vtl[attn_v_off(f, attn_vperm(c), d)] = (_Float16)0.f;
    }
  }
}

// One workgroup = PT token columns x all features.  tp = padded column count.
__global__ __launch_bounds__(256) void k_attn_pack(
    const float* __restrict__ q, int q_stride, const float* __restrict__ k, int k_stride,
    const float* __restrict__ v, int v_stride, const int* __restrict__ cu, int nseg, int d_model,
    int t_total, int tp, const float* __restrict__ scales, _Float16* __restrict__ qh,
    _Float16* __restrict__ ql, _Float16* __restrict__ kh, _Float16* __restrict__ kl,
    _Float16* __restrict__ vth, _Float16* __restrict__ vtl) {
  const float qmul = scales[0], kmul = scales[1], vmul = scales[2];
  __shared__ __align__(16) _Float16 Lh[128 * PS], Ll[128 * PS];
  __shared__ int tok[PT];
  const int tid = threadIdx.x;
  const int col0 = blockIdx.x * PT;
  if (tid < PT) {
    const int c = col0 + tid;
    // last segment whose first column is <= c
    int lo = 0, hi = nseg;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (vstart(cu, mid) <= c) lo = mid; else hi = mid;
    }
    const int j = c - vstart(cu, lo);
    tok[tid] = (j >= 0 && j < cu[lo + 1] - cu[lo]) ? cu[lo] + j : -1;
  }
  __syncthreads();
  const int c4 = (tid & 31) * 4, r0 = tid >> 5;   // 4 features, 8 token rows per pass
  for (int half = 0; half < d_model; half += 128) {
    if (half) __syncthreads();
#pragma unroll 2
    for (int it = 0; it < PT / 8; ++it) {
      const int r = it * 8 + r0;
      const int t = tok[r];
      const int f = half + c4;
      float4 vq = make_float4(0.f, 0.f, 0.f, 0.f), vk = vq, vv = vq;
      if (t >= 0) {
        vq = *reinterpret_cast<const float4*>(q + (size_t)t * q_stride + f);
        vk = *reinterpret_cast<const float4*>(k + (size_t)t * k_stride + f);
        vv = *reinterpret_cast<const float4*>(v + (size_t)t * v_stride + f);
      }
      unsigned int ha, hb, la, lb;
      if (t >= 0) {
        split_pk_s(vq.x, vq.y, qmul, ha, la);
        split_pk_s(vq.z, vq.w, qmul, hb, lb);
        // Q / K planes are head-major [head][token][32]: a 64-key tile of one head
        // is 4 KiB contiguous (full 128-byte lines for the attention kernel's loads)
        const size_t hm = ((size_t)(f / HD) * t_total + t) * HD + f % HD;
        *reinterpret_cast<u32x2*>(qh + hm) = (u32x2){ha, hb};
        *reinterpret_cast<u32x2*>(ql + hm) = (u32x2){la, lb};
        split_pk_s(vk.x, vk.y, kmul, ha, la);
        split_pk_s(vk.z, vk.w, kmul, hb, lb);
        *reinterpret_cast<u32x2*>(kh + hm) = (u32x2){ha, hb};
        *reinterpret_cast<u32x2*>(kl + hm) = (u32x2){la, lb};
      }
      split_pk_s(vv.x, vv.y, vmul, ha, la);
      split_pk_s(vv.z, vv.w, vmul, hb, lb);
      const h16x4 vh4 = __builtin_bit_cast(h16x4, (u32x2){ha, hb});
      const h16x4 vl4 = __builtin_bit_cast(h16x4, (u32x2){la, lb});
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Lh[(c4 + e) * PS + r] = vh4[e];
        Ll[(c4 + e) * PS + r] = vl4[e];
      }
    }
    __syncthreads();
    const int ch = (tid & 7) * 8, d0 = tid >> 3;   // 16-byte chunk, 32 feature rows per pass
    // plane order inside a 16-column group: [0-3, 8-11, 4-7, 12-15] (attn_vperm): the chunk at plane columns ch .. ch + 7
    // holds the tile columns c0 .. c0 + 3 and c0 + 8 .. c0 + 11, c0 = 16 (ch / 16) + 4 * ((ch / 8) & 1)
    const int c0 = (ch & ~15) + ((ch & 8) >> 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int d = it * 32 + d0;
      if (half + d < d_model) {
        const u32x2 a0 = *reinterpret_cast<const u32x2*>(Lh + d * PS + c0), a1 = *reinterpret_cast<const u32x2*>(Lh + d * PS + c0 + 8);
        const u32x2 b0 = *reinterpret_cast<const u32x2*>(Ll + d * PS + c0), b1 = *reinterpret_cast<const u32x2*>(Ll + d * PS + c0 + 8);
        *reinterpret_cast<u32x4*>(vth + attn_v_off(half + d, col0 + ch, d_model)) = (u32x4){a0[0], a0[1], a1[0], a1[1]};
        *reinterpret_cast<u32x4*>(vtl + attn_v_off(half + d, col0 + ch, d_model)) = (u32x4){b0[0], b0[1], b1[0], b1[1]};
      }
    }
  }
}

constexpr int QW2 = 64;    // queries per wave (two 32-query MFMA column blocks) of the default form
constexpr int QB2 = 256;   // queries per workgroup of the default form

// max over the two half-waves (lanes l and l^32) without touching LDS
__device__ __forceinline__ float half_swap_max(float x) {
  const unsigned int u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// H3 = true: split-fp16 (hi + lo planes, 3 MFMAs per product); false: hi planes only.
// NQ = 32-query blocks per wave: 2 (default: 64 queries per wave, 256 per workgroup, ~250 VGPRs, two waves per
// SIMD) or 1 (32 queries per wave, 128 per workgroup, <= 128 VGPRs: four waves per SIMD -- the same instruction
// stream at twice the occupancy; K / V^T fragments are then read from LDS twice as often per query).
template <bool H3, bool LAZY = false, int NQ = 2, int WPS = (NQ == 1 ? 3 : 2), bool PIPE = false>
__global__ __launch_bounds__(256, WPS) void k_attn_h3(
    const _Float16* __restrict__ qh_g, const _Float16* __restrict__ ql_g,
    const _Float16* __restrict__ kh_g, const _Float16* __restrict__ kl_g,
    const _Float16* __restrict__ vth_g, const _Float16* __restrict__ vtl_g, int t_total, int tp,
    const int* __restrict__ cu, const int* __restrict__ kv_seg, int nseg, int nhead,
    const float* __restrict__ scales, float* __restrict__ out, int o_stride, float* __restrict__ lse_out) {
  __shared__ __align__(16) _Float16 Kh[2][KT2 * KH], Kl[2][KT2 * KH];
  __shared__ __align__(16) _Float16 Vth[2][HD * VH], Vtl[2][HD * VH];
  // 1-D grid: all query tiles of one (segment, head) -- which stream the same
  // K/V -- are placed on one XCD (ids b and b+8 share an L2)
  int seg, head, qt;
  {
    const int nqt = gridDim.x / (nhead * nseg);
    const int ngrp = nhead * nseg;
    const int b = blockIdx.x;
    if ((ngrp & 7) == 0) {
      const int xcd = b & 7, idx = b >> 3;
      const int g = xcd + 8 * (idx / nqt);
      qt = idx % nqt;
      head = g % nhead;
      seg = g / nhead;
    } else {
      qt = b % nqt;
      head = (b / nqt) % nhead;
      seg = b / (nqt * nhead);
    }
  }
  const int qbeg = cu[seg], qlen = cu[seg + 1] - qbeg;
  constexpr int QWN = 32 * NQ, QBN = 4 * QWN;   // queries per wave / workgroup
  const int q0 = qt * QBN;
  if (q0 >= qlen) return;
  const int ks = kv_seg[seg];
  const int kbeg = cu[ks], klen = cu[ks + 1] - kbeg;
  const int vbeg = vstart(cu, ks);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int hoff = head * HD;

  // Q^T B-fragments of the wave's two 32-query blocks: lane (query l31, half
  // lh), k-step s: d = 16 s + 8 lh + j
  h16x8 qh[NQ][2], ql[NQ][2];
#pragma unroll
  for (int h = 0; h < NQ; ++h) {
    const int qi = min(q0 + wave * QWN + 32 * h + l31, qlen - 1);
    const size_t row = ((size_t)head * t_total + qbeg + qi) * HD + 8 * lh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qh[h][s] = *reinterpret_cast<const h16x8*>(qh_g + row + 16 * s);
      if constexpr (H3) ql[h][s] = *reinterpret_cast<const h16x8*>(ql_g + row + 16 * s);
    }
  }

  f32x16 o[NQ];
  float m_run[NQ];
  f32x2 psum2[NQ];   // per-lane partial row sums; the two half-waves are joined at the end
  float psum_a[NQ], psum_b[NQ];   // the same for the lazy form (even / odd registers)
  // LAZY: the softmax reference m_ref of the lane's two queries rides into the score MFMAs as
  // their C operand (all 16 registers = 4 - m_ref: scores come out as s - m_ref + 4, ready for
  // exp2), and is moved -- with the accumulator rescale -- only on the first tile and when a score
  // would push a scaled probability out of fp16's range; see tile().
  f32x16 negm[NQ];
#pragma unroll
  for (int h = 0; h < NQ; ++h) {
    m_run[h] = -INFINITY;
    psum2[h] = (f32x2){0.f, 0.f};
    psum_a[h] = psum_b[h] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[h][r] = 0.f;
      negm[h][r] = 0.f;
    }
  }

  // staging roles: K -- key row tid>>2, 16-byte chunk tid&3; V^T -- feature row
  // tid>>3, 16-byte chunk tid&7 (hi and lo planes each)
  const int skr = tid >> 2, skc = (tid & 3) * 8;
  const int svr = tid >> 3, svc = (tid & 7) * 8;
  u32x4 rkh, rkl, rvh, rvl;
  auto fetch = [&](int kt) {
    const size_t krow = ((size_t)head * t_total + kbeg + min(kt + skr, klen - 1)) * HD + skc;
    const size_t vrow = attn_v_off(hoff + svr, (size_t)(vbeg + kt + svc), nhead * HD);
    const _Float16 *a = kh_g + krow, *b = kl_g + krow, *c = vth_g + vrow, *d = vtl_g + vrow;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rkh) : "v"(a));
    if constexpr (H3) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rkl) : "v"(b));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rvh) : "v"(c));
    if constexpr (H3) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rvl) : "v"(d));
  };
  auto stash = [&](int buf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<u32x4*>(Kh[buf] + skr * KH + skc) = rkh;
    if constexpr (H3) *reinterpret_cast<u32x4*>(Kl[buf] + skr * KH + skc) = rkl;
    *reinterpret_cast<u32x4*>(Vth[buf] + svr * VH + svc) = rvh;
    if constexpr (H3) *reinterpret_cast<u32x4*>(Vtl[buf] + svr * VH + svc) = rvl;
  };

  // One 64-key tile.  Branch-free inside (the online-softmax rescale is applied
  // every tile) so that the compiler can run one query block's softmax VALU in
  // the shadow of the other block's MFMAs.
  auto tile = [&](int kt, int buf, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    // K A-fragments (shared by both query blocks; NQ = 1: read per key sub-tile, right before its MFMAs -- the
    // register budget of four waves per SIMD has no room for all eight at once)
    h16x8 kfh[2][2], kfl[2][2];
    auto load_kf = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        kfh[kk][s] = *reinterpret_cast<const h16x8*>(Kh[buf] + (32 * kk + l31) * KH + 16 * s + 8 * lh);
        if constexpr (H3)
          kfl[kk][s] = *reinterpret_cast<const h16x8*>(Kl[buf] + (32 * kk + l31) * KH + 16 * s + 8 * lh);
      }
    };
    if constexpr (NQ == 2) {
      load_kf(0);
      load_kf(1);
    }
    // ---- S^T = K Q^T (rows = keys, cols = queries) ----
    f32x16 sacc[NQ][2];
#pragma unroll
    for (int h = 0; h < NQ; ++h)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if constexpr (NQ == 1) {
          if (kk == 1) __builtin_amdgcn_sched_barrier(0);
          load_kf(kk);
        }
        if constexpr (LAZY) {
          sacc[h][kk] = negm[h];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[h][kk][r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if constexpr (H3) {
            sacc[h][kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[kk][s], ql[h][s], sacc[h][kk], 0, 0, 0);
            sacc[h][kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[kk][s], qh[h][s], sacc[h][kk], 0, 0, 0);
          }
          sacc[h][kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[kk][s], qh[h][s], sacc[h][kk], 0, 0, 0);
        }
      }
    // V^T A-fragments (shared by both query blocks; NQ = 1: read per key sub-tile in front of its MFMAs)
    h16x8 vfh[2][2], vfl[2][2];
    auto load_vf = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        // (round 5: the planes hold every 16-key group in fragment order, attn_vperm: one 16-byte read)
        vfh[kk][s] = *reinterpret_cast<const h16x8*>(Vth[buf] + l31 * VH + 32 * kk + 16 * s + 8 * lh);
        if constexpr (H3) vfl[kk][s] = *reinterpret_cast<const h16x8*>(Vtl[buf] + l31 * VH + 32 * kk + 16 * s + 8 * lh);
      }
    };
    if constexpr (NQ == 2) {
      load_vf(0);
      load_vf(1);
    }
#pragma unroll
    for (int h = 0; h < NQ; ++h) {
      // ---- online softmax (lane = query; reg r of sub-tile kk <-> key
      //      32 kk + (r&3) + 8 (r>>2) + 4 lh) ----
      if (TAIL) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = kt + 32 * kk + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (j >= klen) sacc[h][kk][r] = -INFINITY;
          }
      }
      float mx = fmaxf(sacc[h][0][0], sacc[h][1][0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(sacc[h][0][r], sacc[h][1][r]));
      mx = half_swap_max(mx);
      if constexpr (LAZY) {
        // sacc = s - m_ref + kLazyOff.  Recentre (first tile: always; later: some query of the wave
        // has a score more than 2^(15.5 - kLazyOff) above its reference -- the scaled probability
        // would leave fp16's range): delta moves every lane's reference to its current maximum, like
        // the eager form does on every tile.  kLazyOff = 4 (the eager form uses 10): probabilities
        // below 2^-18 of the row maximum go subnormal in the hi plane, an absolute error of
        // 2^-29 of the maximum -- and a reference has 11.5 octaves of headroom before it must move.
        constexpr float kLazyOff = 4.0f;
        const bool first = kt == 0;
        if (first || __builtin_amdgcn_ballot_w64(mx > 15.5f) != 0) {
          const float delta = first ? mx - kLazyOff : fmaxf(mx - kLazyOff, 0.f);
          const float corr = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // o, psum are 0 on the first tile
          psum_a[h] *= corr;
          psum_b[h] *= corr;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            o[h][r] *= corr;
            negm[h][r] -= delta;
            sacc[h][0][r] -= delta;
            sacc[h][1][r] -= delta;
          }
        }
        unsigned int ph_u[2][8], pl_u[2][8];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 pv;
            pv[0] = __builtin_amdgcn_exp2f(sacc[h][kk][r]);
            pv[1] = __builtin_amdgcn_exp2f(sacc[h][kk][r + 1]);
            psum_a[h] += pv[0];   // two plain adds (the file is built with -fno-slp-vectorize): v_pk_add_f32 issues
            psum_b[h] += pv[1];   // slower than the pair beside MFMAs (MI355X_MICROARCH price list)
            const unsigned int hi_u =
                __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(pv[0], pv[1]));
            ph_u[kk][r >> 1] = hi_u;
            if constexpr (H3) {
              unsigned int lo_u;
              asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                  "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                  : "=&v"(lo_u)
                  : "v"(hi_u), "v"(pv[0]), "v"(pv[1]));
              pl_u[kk][r >> 1] = lo_u;
            }
          }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          if constexpr (NQ == 1) {
            if (kk == 1) __builtin_amdgcn_sched_barrier(0);
            load_vf(kk);
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const u32x4 pa = {ph_u[kk][4 * s], ph_u[kk][4 * s + 1], ph_u[kk][4 * s + 2], ph_u[kk][4 * s + 3]};
            const h16x8 pbh = __builtin_bit_cast(h16x8, pa);
            if constexpr (H3) {
              const u32x4 pb = {pl_u[kk][4 * s], pl_u[kk][4 * s + 1], pl_u[kk][4 * s + 2], pl_u[kk][4 * s + 3]};
              const h16x8 pbl = __builtin_bit_cast(h16x8, pb);
              o[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[kk][s], pbl, o[h], 0, 0, 0);
              o[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[kk][s], pbh, o[h], 0, 0, 0);
            }
            o[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[kk][s], pbh, o[h], 0, 0, 0);
          }
        }
        continue;
      }
      const float m_new = fmaxf(m_run[h], mx);
      const float corr = __builtin_amdgcn_exp2f(m_run[h] - m_new);   // first tile: exp2(-inf) = 0
      m_run[h] = m_new;
      const f32x2 c2 = {corr, corr};
      psum2[h] *= c2;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 t = {o[h][r], o[h][r + 1]};
        t *= c2;
        o[h][r] = t[0];
        o[h][r + 1] = t[1];
      }
      // p' = 2^10 * exp2(s - m): the 2^10 keeps small probabilities inside fp16's
      // normal range; the row sum accumulates the same scaled values, so it
      // cancels in the final 1/l.
      const float mshift = m_new - 10.0f;
      const f32x2 ms2 = {mshift, mshift};
      unsigned int ph_u[2][8], pl_u[2][8];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 sv = {sacc[h][kk][r], sacc[h][kk][r + 1]};
          const f32x2 dv = sv - ms2;                       // v_pk_add_f32
          f32x2 pv;
          pv[0] = __builtin_amdgcn_exp2f(dv[0]);
          pv[1] = __builtin_amdgcn_exp2f(dv[1]);
          psum2[h] += pv;                                  // v_pk_add_f32
          const unsigned int hi_u =
              __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(pv[0], pv[1]));
          ph_u[kk][r >> 1] = hi_u;
          if constexpr (H3) {
            unsigned int lo_u;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "=&v"(lo_u)
                : "v"(hi_u), "v"(pv[0]), "v"(pv[1]));
            pl_u[kk][r >> 1] = lo_u;
          }
        }
      // ---- O^T += V^T P^T ----
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const u32x4 pa = {ph_u[kk][4 * s], ph_u[kk][4 * s + 1], ph_u[kk][4 * s + 2], ph_u[kk][4 * s + 3]};
          const h16x8 pbh = __builtin_bit_cast(h16x8, pa);
          if constexpr (H3) {
            const u32x4 pb = {pl_u[kk][4 * s], pl_u[kk][4 * s + 1], pl_u[kk][4 * s + 2], pl_u[kk][4 * s + 3]};
            const h16x8 pbl = __builtin_bit_cast(h16x8, pb);
            o[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[kk][s], pbl, o[h], 0, 0, 0);
            o[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[kk][s], pbh, o[h], 0, 0, 0);
          }
          o[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[kk][s], pbh, o[h], 0, 0, 0);
        }
    }
  };

  // PIPE (NQ = 1, lazy softmax): the same tile as two 32-key halves whose phases are staggered inside the wave --
  // the score MFMAs of half 1 are independent of the exponentials of half 0, and the PV MFMAs of half 0 of the
  // exponentials of half 1, so 12 of the tile's 24 MFMAs have vector work of the same wave to run beside.  The lazy
  // reference is checked per half; a recentring at half 1 finds the probabilities of half 0 already converted: its
  // shift is rounded up to an integer so that they can be rescaled exactly (a power of two) in their fp16 planes.
  auto tile_pipe = [&](int kt, int buf, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    constexpr float kLazyOff = 4.0f;
    h16x8 kfh[2], kfl[2], vfh[2], vfl[2];
    auto load_kf = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        kfh[s] = *reinterpret_cast<const h16x8*>(Kh[buf] + (32 * kk + l31) * KH + 16 * s + 8 * lh);
        if constexpr (H3) kfl[s] = *reinterpret_cast<const h16x8*>(Kl[buf] + (32 * kk + l31) * KH + 16 * s + 8 * lh);
      }
    };
    auto load_vf = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        vfh[s] = *reinterpret_cast<const h16x8*>(Vth[buf] + l31 * VH + 32 * kk + 16 * s + 8 * lh);
        if constexpr (H3) vfl[s] = *reinterpret_cast<const h16x8*>(Vtl[buf] + l31 * VH + 32 * kk + 16 * s + 8 * lh);
      }
    };
    auto scores = [&](f32x16& acc) __attribute__((always_inline)) {
      acc = negm[0];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if constexpr (H3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], ql[0][s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[s], qh[0][s], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], qh[0][s], acc, 0, 0, 0);
      }
    };
    auto rowmax = [&](f32x16& acc, int kk) __attribute__((always_inline)) {
      if (TAIL) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = kt + 32 * kk + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (j >= klen) acc[r] = -INFINITY;
        }
      }
      float mx = fmaxf(acc[0], acc[1]);
#pragma unroll
      for (int r = 2; r < 16; ++r) mx = fmaxf(mx, acc[r]);
      return half_swap_max(mx);
    };
    auto convert = [&](const f32x16& acc, unsigned int (&ph)[8], unsigned int (&pl)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(acc[r]), p1 = __builtin_amdgcn_exp2f(acc[r + 1]);
        psum_a[0] += p0;
        psum_b[0] += p1;
        const unsigned int hi_u = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(p0, p1));
        ph[r >> 1] = hi_u;
        if constexpr (H3) {
          unsigned int lo_u;
          asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
              "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
              : "=&v"(lo_u)
              : "v"(hi_u), "v"(p0), "v"(p1));
          pl[r >> 1] = lo_u;
        }
      }
    };
    auto pv = [&](const unsigned int (&ph)[8], const unsigned int (&pl)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4 pa = {ph[4 * s], ph[4 * s + 1], ph[4 * s + 2], ph[4 * s + 3]};
        const h16x8 pbh = __builtin_bit_cast(h16x8, pa);
        if constexpr (H3) {
          const u32x4 pb = {pl[4 * s], pl[4 * s + 1], pl[4 * s + 2], pl[4 * s + 3]};
          const h16x8 pbl = __builtin_bit_cast(h16x8, pb);
          o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[s], pbl, o[0], 0, 0, 0);
          o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[s], pbh, o[0], 0, 0, 0);
        }
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[s], pbh, o[0], 0, 0, 0);
      }
    };
    // ---- half 0: scores, reference check ----
    f32x16 s0, s1;
    load_kf(0);
    scores(s0);
    const float mx0 = rowmax(s0, 0);
    const bool first = kt == 0;
    if (first || __builtin_amdgcn_ballot_w64(mx0 > 15.5f) != 0) {
      const float delta = first ? mx0 - kLazyOff : fmaxf(mx0 - kLazyOff, 0.f);
      const float corr = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);   // o, psum are 0 on the first tile
      psum_a[0] *= corr;
      psum_b[0] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= corr;
        negm[0][r] -= delta;
        s0[r] -= delta;
      }
    }
    // ---- scores of half 1 beside the exponentials of half 0 ----
    unsigned int ph0[8], pl0[8], ph1[8], pl1[8];
    load_kf(1);
    scores(s1);
    convert(s0, ph0, pl0);
    const float mx1 = rowmax(s1, 1);
    if (__builtin_amdgcn_ballot_w64(mx1 > 15.5f) != 0) {
      const float delta = ceilf(fmaxf(mx1 - kLazyOff, 0.f));          // integer: corr is an exact power of two
      const float corr = __builtin_amdgcn_exp2f(-delta);
      psum_a[0] *= corr;
      psum_b[0] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] *= corr;
        negm[0][r] -= delta;
        s1[r] -= delta;
      }
      typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
      const _Float16 ch = (_Float16)corr;                              // below 2^-24: the old probabilities vanish
      const h16x2 c2 = {ch, ch};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        ph0[i] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(h16x2, ph0[i]) * c2);
        if constexpr (H3) pl0[i] = __builtin_bit_cast(unsigned int, __builtin_bit_cast(h16x2, pl0[i]) * c2);
      }
    }
    // ---- PV of half 0 beside the exponentials of half 1, then PV of half 1 ----
    load_vf(0);
    pv(ph0, pl0);
    convert(s1, ph1, pl1);
    load_vf(1);
    pv(ph1, pl1);
  };

  if (klen > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();

  int buf = 0, kt = 0;
  for (; kt + KT2 <= klen; kt += KT2, buf ^= 1) {
    const bool more = kt + KT2 < klen;
    if (more) fetch(kt + KT2);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PIPE) tile_pipe(kt, buf, std::false_type{});
    else tile(kt, buf, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  if (kt < klen) {
    if constexpr (PIPE) tile_pipe(kt, buf, std::true_type{});
    else tile(kt, buf, std::true_type{});
  }

#pragma unroll
  for (int h = 0; h < NQ; ++h) {
    float l_run = LAZY ? psum_a[h] + psum_b[h] : psum2[h][0] + psum2[h][1];
    l_run += __shfl_xor(l_run, 32, 64);
    const int qi = q0 + wave * QWN + 32 * h + l31;
    if constexpr (LAZY) {
      // log2 sum_j 2^(s_ij) of the query's scores (s in the kernel's base-2 units = log2(e) q.k / sqrt(d)): the
      // probabilities above are 2^(s - m_ref + off) with negm = off - m_ref.  Handed to the backward (training).
      if (lse_out != nullptr && qi < qlen && lh == 0)
        lse_out[(size_t)(qbeg + qi) * nhead + head] = __builtin_amdgcn_logf(l_run) - negm[h][0];
    }
    if (qi < qlen) {
      const float inv = (l_run > 0.f ? 1.0f / l_run : 0.f) * scales[3];   // 2^-ev undoes the V multiplier
      float* op = out + (size_t)(qbeg + qi) * o_stride + hoff;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 w4;
        w4.x = o[h][4 * g + 0] * inv;
        w4.y = o[h][4 * g + 1] * inv;
        w4.z = o[h][4 * g + 2] * inv;
        w4.w = o[h][4 * g + 3] * inv;
        *reinterpret_cast<float4*>(op + 8 * g + 4 * lh) = w4;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Round 5 core ("s" = sum-checked).  Same contract, operand planes, LDS images and staging as
// k_attn_h3<H3, /*LAZY*/ true, /*NQ*/ 1>, with two changes that take vector instructions off the per-score path
// (the kernel is bound by vector-instruction issue: profiles/r04_attn_counters.txt):
//  * no row-maximum pass in the steady state.  The lazy reference is set from the true maximum on the first tile;
//    afterwards a tile is exponentiated optimistically and the row sums the softmax needs anyway are the check: a
//    lane's 32 probabilities of the tile are all below their sum, so "sum < 2^15" proves that nothing left fp16's
//    range.  Otherwise (rare: a score more than ~10 octaves above the reference) the tile is recomputed after
//    recentring on its true maximum -- the loop below runs its body a second time, nothing of the tile has been
//    accumulated yet.  24 v_max + swap + compare per tile become one add and one compare.
//  * PLO = false ("single probability plane", attention mode 3): P is carried as ONE fp16 plane (round toward
//    zero) instead of hi + lo, O^T = (V_hi + V_lo)^T P_hi: 8 instead of 12 MFMAs for the second product and no
//    v_fma_mix pair per probability.  The row sum is formed from the ROUNDED plane (v_dot2c_f32_f16 with a ones
//    vector), so numerator and denominator use the same weights: the output is an exact convex combination of the
//    value rows with weights perturbed by < 2^-10 relative each; the systematic part of the rounding cancels in
//    1 / l, what is left is sum_j w_j delta_j (v_j - o), delta uniform over one fp16 ulp.
//    Q K^T keeps the three-product split in every H3 mode: score errors are amplified by the exponential.
// PRIO > 0: s_setprio PRIO around the score MFMAs (A/B switch SPR_ATTN_PRIO).
// ABL (diagnostic builds only, -DSPR_ATTN_ABLATE; results are garbage): 1 = no fragment reads from LDS (the Q registers
// stand in), 2 = no K / V staging (no global loads, no LDS writes), 4 = no per-tile barrier.
// ADAPT (attention mode 4, needs H3 && PLO): the lo plane of the probabilities only where it matters.  A tile is
// "significant" for a wave when some lane holds a probability of at least 2^-7 of that lane's running row sum (the
// maximum of the tile's hi plane: 15 v_pk_max_f16); only such tiles pay for the lo plane (32 quarter-rate
// v_fma_mix*_f16), the f32 sums and the third MFMA of every k-step.  The other tiles run as mode 3 (row sum from the
// rounded plane).  Every key with weight w_j >= 2^-7 is then carried with full accuracy; what the rest can contribute is
// bounded by 2^-10 sqrt(sum w_j^2) <= 2^-10 sqrt(2^-7) = 8.6e-5 of the value spread in the worst case (128 keys
// of exactly that weight) and is ~5e-6 for flat rows of 2 000 keys -- the peaked rows mode 3 is inaccurate on
// (few keys with large weights) are exact here.
// PF (round 5): fragment prefetch.  The sensitivity builds (profiles/r05_attn_ablate.txt) price the LDS fragment
// reads at a quarter of the kernel: every group of MFMAs waits for four ds_read_b128 issued right in front of it.
// PF = 1 issues a tile's V^T fragments (both 32-key halves) at the top of the tile -- they land under the score
// MFMAs and the exponentials -- and the K fragments of both halves before the first score MFMA.
template <bool H3, bool PLO, int WPS, int PRIO, int ABL = 0, bool ADAPT = false, int PF = 0>
__global__ __launch_bounds__(256, WPS) void k_attn_s(
    const _Float16* __restrict__ qh_g, const _Float16* __restrict__ ql_g,
    const _Float16* __restrict__ kh_g, const _Float16* __restrict__ kl_g,
    const _Float16* __restrict__ vth_g, const _Float16* __restrict__ vtl_g, int t_total, int tp,
    const int* __restrict__ cu, const int* __restrict__ kv_seg, int nseg, int nhead,
    const float* __restrict__ scales, float* __restrict__ out, int o_stride, float* __restrict__ lse_out,
    const int* __restrict__ o_tiles, float sig_thr) {
  static_assert(H3 || !PLO, "a lo plane of P only with split operands");
  static_assert(!ADAPT || (H3 && PLO), "the adaptive form is a refinement of the split form");
  // K tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): the image is lane-linear,
  // 64 key rows of 64 bytes; the 16-byte chunk c of row r sits at position c ^ ((r >> 2) & 3) (the DMA lane fetches the
  // SOURCE chunk of its position: linear destination + swizzled source + the same XOR on the read), which makes the
  // fragment reads (16 lanes = 16 rows, one chunk) conflict-free without row padding.
  // V^T tiles the same way: 32 feature rows of 128 bytes (64 keys in the planes' fragment order, attn_vperm); chunk c of
  // row d at position c ^ ((d >> 1) & 7).
  __shared__ __align__(16) _Float16 Kh[2][KT2 * HD], Kl[2][KT2 * HD];
  __shared__ __align__(16) _Float16 Vth[2][HD * KT2], Vtl[2][HD * KT2];
  int seg, head, qt;
  {
    const int nqt = gridDim.x / (nhead * nseg);
    const int ngrp = nhead * nseg;
    const int b = blockIdx.x;
    if ((ngrp & 7) == 0) {      // all query tiles of one (segment, head) on one XCD (ids b and b + 8 share an L2)
      const int xcd = b & 7, idx = b >> 3;
      const int g = xcd + 8 * (idx / nqt);
      qt = idx % nqt;
      head = g % nhead;
      seg = g / nhead;
    } else {
      qt = b % nqt;
      head = (b / nqt) % nhead;
      seg = b / (nqt * nhead);
    }
  }
  const int qbeg = cu[seg], qlen = cu[seg + 1] - qbeg;
  const int q0 = qt * 128;
  if (q0 >= qlen) return;
  const int ks = kv_seg[seg];
  const int kbeg = cu[ks], klen = cu[ks + 1] - kbeg;
  const int vbeg = vstart(cu, ks);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int hoff = head * HD;

  h16x8 qh[2], ql[2];
  {
    const int qi = min(q0 + wave * 32 + l31, qlen - 1);
    const size_t row = ((size_t)head * t_total + qbeg + qi) * HD + 8 * lh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qh[s] = *reinterpret_cast<const h16x8*>(qh_g + row + 16 * s);
      if constexpr (H3) ql[s] = *reinterpret_cast<const h16x8*>(ql_g + row + 16 * s);
    }
  }
  // The Q fragments are first USED inside the tile loop: without this the compiler's own wait for their loads
  // lands in the loop (s_waitcnt vmcnt(0) in front of the score MFMAs) and drains the asm-issued prefetch of the
  // next K / V tile on every iteration.
  if constexpr (H3) asm volatile("" ::"v"(qh[0]), "v"(qh[1]), "v"(ql[0]), "v"(ql[1]));
  else asm volatile("" ::"v"(qh[0]), "v"(qh[1]));
  f32x16 o, negm;      // negm: all registers = off - m_ref, the C operand of the score MFMAs
  float psum = 0.f;    // per-lane partial row sum (the two half-waves are joined at the end)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    o[r] = 0.f;
    negm[r] = 0.f;
  }

  const int skr = tid >> 2, skc = ((tid & 3) ^ ((tid >> 4) & 3)) * 8;   // K: row, SOURCE chunk of this lane's position
  const int svr = tid >> 3, svc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;   // V^T: row, SOURCE chunk of this lane's position
  const unsigned kdst_h = (unsigned)(uintptr_t)(&Kh[0][0]) + (unsigned)wave * 1024u;   // + buf * 4096
  const unsigned kdst_l = (unsigned)(uintptr_t)(&Kl[0][0]) + (unsigned)wave * 1024u;
  const unsigned vdst_h = (unsigned)(uintptr_t)(&Vth[0][0]) + (unsigned)wave * 1024u;
  const unsigned vdst_l = (unsigned)(uintptr_t)(&Vtl[0][0]) + (unsigned)wave * 1024u;
  auto dma16 = [&](const void* base, unsigned off, unsigned dst) __attribute__((always_inline)) {
    const unsigned d = __builtin_amdgcn_readfirstlane(dst);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(d)
                 : "memory", "m0");
  };
  auto fetch = [&](int kt, int nbuf) {     // nbuf: the buffer tile kt will be read from
    if constexpr (ABL & 2) return;
    const unsigned koff = (unsigned)((((size_t)head * t_total + kbeg + min(kt + skr, klen - 1)) * HD + skc) * 2);
    const unsigned voff = (unsigned)(attn_v_off(hoff + svr, (size_t)(vbeg + kt + svc), nhead * HD) * 2);
    dma16(kh_g, koff, kdst_h + (unsigned)nbuf * 4096u);
    if constexpr (H3) dma16(kl_g, koff, kdst_l + (unsigned)nbuf * 4096u);
    dma16(vth_g, voff, vdst_h + (unsigned)nbuf * 4096u);
    if constexpr (H3) dma16(vtl_g, voff, vdst_l + (unsigned)nbuf * 4096u);
    if constexpr (ABL & 16) {   // sensitivity: the same four pieces once more
      dma16(kh_g, koff, kdst_h + (unsigned)nbuf * 4096u);
      dma16(kl_g, koff, kdst_l + (unsigned)nbuf * 4096u);
      dma16(vth_g, voff, vdst_h + (unsigned)nbuf * 4096u);
      dma16(vtl_g, voff, vdst_l + (unsigned)nbuf * 4096u);
    }
  };
  auto stash = [&](int) {
    if constexpr (ABL & 2) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next tile have landed
  };

  constexpr float kOff = 4.0f;        // a fresh reference puts the row maximum at 2^kOff
  constexpr float kSumMax = 32768.f;  // a lane's tile sum below this: every probability inside fp16's range
  const float kSig = sig_thr;         // ADAPT: a probability below kSig x the lane's running sum needs no lo plane

  auto tile = [&](int kt, int buf, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    f32x16 sc[2];
    unsigned int ph_u[2][8], pl_u[2][8];
    float ts = 0.f;
    bool redo = false;
    h16x8 pvh[2][2], pvl[2][2];     // PF: V^T fragments of the whole tile
    if constexpr (PF > 0) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int vpos = l31 * KT2 + 8 * ((4 * kk + 2 * s + lh) ^ ((l31 >> 1) & 7));
          pvh[kk][s] = *reinterpret_cast<const h16x8*>(Vth[buf] + vpos);
          if constexpr (H3) pvl[kk][s] = *reinterpret_cast<const h16x8*>(Vtl[buf] + vpos);
        }
    }
    bool sigk[2] = {PLO, PLO};      // does the 32-key sub-tile carry a lo plane (ADAPT: decided per sub-tile)
    for (;;) {
      // ---- S^T = K Q^T + (off - m_ref)   (rows = keys, cols = queries) ----
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
      h16x8 kfh[2][2], kfl[2][2];
      auto load_kf = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if constexpr (ABL & 1) {
            kfh[kk][s] = qh[s];
            if constexpr (H3) kfl[kk][s] = ql[s];
            continue;
          }
          const int kpos = (32 * kk + l31) * HD + 8 * ((2 * s + lh) ^ ((l31 >> 2) & 3));
          kfh[kk][s] = *reinterpret_cast<const h16x8*>(Kh[buf] + kpos);
          if constexpr (H3) kfl[kk][s] = *reinterpret_cast<const h16x8*>(Kl[buf] + kpos);
          if constexpr (ABL & 8) {    // sensitivity: the same reads once more (volatile: not merged), data unchanged
            const h16x8 d0 = *reinterpret_cast<const volatile h16x8*>(Kh[buf ^ 1] + kpos);
            const h16x8 d1 = *reinterpret_cast<const volatile h16x8*>(Kl[buf ^ 1] + kpos);
            asm volatile("" ::"v"(d0), "v"(d1));
          }
        }
      };
      if constexpr (PF > 0) {
        load_kf(0);
        load_kf(1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if constexpr (PF == 0) load_kf(kk);
        sc[kk] = negm;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if constexpr (H3) {
            sc[kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[kk][s], ql[s], sc[kk], 0, 0, 0);
            sc[kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[kk][s], qh[s], sc[kk], 0, 0, 0);
          }
          sc[kk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[kk][s], qh[s], sc[kk], 0, 0, 0);
        }
        if (kk == 0) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(0);
      if (TAIL) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = kt + 32 * kk + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (j >= klen) sc[kk][r] = -INFINITY;
          }
      }
      const bool first = kt == 0;
      if (first || redo) {
        // recentre on the tile's true maximum (first tile: o and psum are still zero)
        float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(sc[0][r], sc[1][r]));
        mx = half_swap_max(mx);
        const float delta = first ? mx - kOff : fmaxf(mx - kOff, 0.f);
        const float corr = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
        psum *= corr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          o[r] *= corr;
          negm[r] -= delta;
          sc[0][r] -= delta;
          sc[1][r] -= delta;
        }
      }
      // ---- probabilities, their planes and the lane's tile sum ----
      float ta = 0.f, tb = 0.f;
      if constexpr (ADAPT) {
        // per 32-key sub-tile (the decision needs that sub-tile's probabilities only: 16 live registers, not 32)
        typedef _Float16 h16x2_ __attribute__((ext_vector_type(2)));
        const float thr = kSig * psum;                                    // (first tile: psum = 0 -> significant)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          h16x2_ m2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float p0 = __builtin_amdgcn_exp2f(sc[kk][r]);
            const float p1 = __builtin_amdgcn_exp2f(sc[kk][r + 1]);
            sc[kk][r] = p0;
            sc[kk][r + 1] = p1;
            // round to NEAREST here (v_cvt_pk_f16_f32, gfx950): sub-tiles with and without a lo plane are mixed in
            // one row, so the systematic part of a round-toward-zero plane (-2^-11.5 on average) would no longer
            // cancel in 1 / l as it does when EVERY weight carries it (mode 3) -- measured 4e-5 of the output scale
            // on the late-spike test before this
            // (through the compiler, not inline asm: a VALU instruction that reads the result of a transcendental needs
            // a wait state the hazard pass only inserts for instructions it can see -- the asm form produced NaNs in
            // the lanes the quarter-rate unit finishes last)
            const unsigned int hi_u = __builtin_bit_cast(unsigned int, __builtin_convertvector((f32x2){p0, p1}, h16x2_));
            ph_u[kk][r >> 1] = hi_u;
            m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(h16x2_, hi_u));
          }
          const float mxp = (float)(m2[0] > m2[1] ? m2[0] : m2[1]);
          sigk[kk] = __builtin_amdgcn_ballot_w64(!(mxp < thr)) != 0;
          if (sigk[kk]) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              ta += sc[kk][r];
              tb += sc[kk][r + 1];
              unsigned int lo_u;
              asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                  "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                  : "=&v"(lo_u)
                  : "v"(ph_u[kk][r >> 1]), "v"(sc[kk][r]), "v"(sc[kk][r + 1]));
              pl_u[kk][r >> 1] = lo_u;
            }
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(ta) : "v"(ph_u[kk][i]));
              asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(tb) : "v"(ph_u[kk][i]));
            }
          }
        }
      } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float p0 = __builtin_amdgcn_exp2f(sc[kk][r]);
          const float p1 = __builtin_amdgcn_exp2f(sc[kk][r + 1]);
          if constexpr (ABL & 64) {   // sensitivity: two more exponentials per pair, results discarded
            float e0, e1;
            asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=v"(e0), "=v"(e1) : "v"(sc[kk][r]), "v"(sc[kk][r + 1]));
            asm volatile("" ::"v"(e0), "v"(e1));
          }
          const auto hi_h = __builtin_amdgcn_cvt_pkrtz(p0, p1);
          const unsigned int hi_u = __builtin_bit_cast(unsigned int, hi_h);
          ph_u[kk][r >> 1] = hi_u;
          if constexpr (PLO) {
            ta += p0;      // two plain adds (-fno-slp-vectorize): v_pk_add_f32 issues slower than the pair
            tb += p1;
            unsigned int lo_u;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
                "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "=&v"(lo_u)
                : "v"(hi_u), "v"(p0), "v"(p1));
            pl_u[kk][r >> 1] = lo_u;
          } else {
            // the sum of the ROUNDED pair: acc += float(half) exactly, one v_fma_mix_f32 per half (a v_dot2c_f32_f16
            // per pair costs 2.7 ns beside an MFMA, these 0.4 each: scripts/abl/issue_probe.hip, round 5)
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(ta) : "v"(hi_u));
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(tb) : "v"(hi_u));
          }
        }
      }
      ts = ta + tb;
      if (redo || __builtin_amdgcn_ballot_w64(!(ts < kSumMax)) == 0) break;
      redo = true;   // some probability may have left fp16's range: once more, recentred (rare)
    }
    psum += ts;
    // ---- O^T += V^T P^T ----
    auto pv = [&](auto lo_tag, int kk) __attribute__((always_inline)) {
      constexpr bool LO = decltype(lo_tag)::value;
      h16x8 vfh[2], vfl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if constexpr (ABL & 1) {
          vfh[s] = qh[s];
          if constexpr (H3) vfl[s] = ql[s];
          continue;
        }
        if constexpr (PF > 0) {
          vfh[s] = pvh[kk][s];
          if constexpr (H3) vfl[s] = pvl[kk][s];
          continue;
        }
        const int vpos = l31 * KT2 + 8 * ((4 * kk + 2 * s + lh) ^ ((l31 >> 1) & 7));
        vfh[s] = *reinterpret_cast<const h16x8*>(Vth[buf] + vpos);
        if constexpr (H3) vfl[s] = *reinterpret_cast<const h16x8*>(Vtl[buf] + vpos);
        if constexpr (ABL & 8) {
          const h16x8 d0 = *reinterpret_cast<const volatile h16x8*>(Vth[buf ^ 1] + vpos);
          const h16x8 d1 = *reinterpret_cast<const volatile h16x8*>(Vtl[buf ^ 1] + vpos);
          asm volatile("" ::"v"(d0), "v"(d1));
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32x4 pa = {ph_u[kk][4 * s], ph_u[kk][4 * s + 1], ph_u[kk][4 * s + 2], ph_u[kk][4 * s + 3]};
        const h16x8 pbh = __builtin_bit_cast(h16x8, pa);
        if constexpr (LO) {
          const u32x4 pb = {pl_u[kk][4 * s], pl_u[kk][4 * s + 1], pl_u[kk][4 * s + 2], pl_u[kk][4 * s + 3]};
          const h16x8 pbl = __builtin_bit_cast(h16x8, pb);
          o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[s], pbl, o, 0, 0, 0);
        }
        if constexpr (H3) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[s], pbh, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[s], pbh, o, 0, 0, 0);
      }
    };
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk == 1) __builtin_amdgcn_sched_barrier(0);
      if constexpr (ADAPT) {
        if (sigk[kk]) pv(std::true_type{}, kk);
        else pv(std::false_type{}, kk);
      } else {
        pv(std::integral_constant<bool, PLO>{}, kk);
      }
    }
  };

  if (klen > 0) {
    fetch(0, 0);
    stash(0);
  }
  __syncthreads();
  int buf = 0, kt = 0;
  for (; kt + KT2 <= klen; kt += KT2, buf ^= 1) {
    const bool more = kt + KT2 < klen;
    // (the barrier at the end of the previous iteration: every wave has finished reading buffer buf ^ 1)
    if (more) fetch(kt + KT2, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    tile(kt, buf, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    if (more) stash(buf ^ 1);
    if constexpr (!(ABL & 4)) __syncthreads();
    if constexpr (ABL & 32) __syncthreads();
  }
  if (kt < klen) tile(kt, buf, std::true_type{});

  float l_run = psum;
  l_run += __shfl_xor(l_run, 32, 64);
  const int qi = q0 + wave * 32 + l31;
  // log2 sum_j 2^(s_ij): handed to the backward (training), as k_attn_h3's lazy form does
  if (lse_out != nullptr && qi < qlen && lh == 0)
    lse_out[(size_t)(qbeg + qi) * nhead + head] = __builtin_amdgcn_logf(l_run) - negm[0];
  if (o_tiles != nullptr) {
    // Tiled output for the fused row chains (xenc.hip, "tiled token tensors"): this workgroup's 128 queries are chain tile
    // o_tiles[seg] + qt, the wave's 32 queries one 32 KiB wave block of it, and the lane's four features 8 g + 4 lh .. + 3
    // of head `head` are the 16-byte piece (4 head + g, lane) of that block -- what the chain's lane (token l31, half lh)
    // loads with ONE coalesced instruction per piece.  Every lane stores (rows past the cloud's end hold the clamped last
    // query's finite values: the chain computes on them and drops them).
    const float inv = (l_run > 0.f ? 1.0f / l_run : 0.f) * scales[3];
    float4* op = reinterpret_cast<float4*>(out) + ((size_t)(o_tiles[seg] + qt) * 4 + wave) * 2048 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w4;
      w4.x = o[4 * g + 0] * inv;
      w4.y = o[4 * g + 1] * inv;
      w4.z = o[4 * g + 2] * inv;
      w4.w = o[4 * g + 3] * inv;
      op[(4 * head + g) * 64] = w4;
    }
  } else if (qi < qlen) {
    const float inv = (l_run > 0.f ? 1.0f / l_run : 0.f) * scales[3];   // 2^-ev undoes the V multiplier
    float* op = out + (size_t)(qbeg + qi) * o_stride + hoff;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 w4;
      w4.x = o[4 * g + 0] * inv;
      w4.y = o[4 * g + 1] * inv;
      w4.z = o[4 * g + 2] * inv;
      w4.w = o[4 * g + 3] * inv;
      *reinterpret_cast<float4*>(op + 8 * g + 4 * lh) = w4;
    }
  }
}

static std::atomic<int> g_attn_mode{4};   // 4 = adaptive lo plane (default), 1 = split-fp16, 0 = exact f32 MFMA, 2 = single-pass fp16,
                                          // 3 = split-fp16 scores, ONE probability plane (k_attn_s<true, false>), 4 = adaptive lo plane

// L1 norm of every row of w [rows, cols]: one wave per row.
__global__ void k_row_l1(const float* __restrict__ w, int rows, int cols, float* __restrict__ out) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += fabsf(w[(size_t)row * cols + c]);
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// Plane multipliers from bounds on |q|, |k|, |v| (one workgroup).
//   rowl1 == nullptr: the bounds are the measured maxima (q_parts, k_parts, v_parts);
//   else (fused in-projection, weights [3d, d], bias [3d]): bound of block b =
//     max|x| * max_row L1(W_b) + max|bias_b|, x = x_qk for Q and K, x_v for V.
__global__ __launch_bounds__(256) void k_plane_scales(const float* __restrict__ q_parts,
                                                      const float* __restrict__ k_parts,
                                                      const float* __restrict__ v_parts,
                                                      const float* __restrict__ rowl1,
                                                      const float* __restrict__ bias, int d, float qscale,
                                                      float* __restrict__ scales, int nq_parts, int nk_parts,
                                                      int nv_parts, float* __restrict__ out_range) {
  __shared__ float sh[17];
  __shared__ float blk[6];
  float qb = block_absmax(q_parts, sh, nq_parts);
  float kb = k_parts == q_parts ? qb : block_absmax(k_parts, sh, nk_parts);   // shared input: reduce once
  float vb = v_parts == q_parts ? qb : block_absmax(v_parts, sh, nv_parts);
  if (rowl1 != nullptr) {
    for (int b = 0; b < 3; ++b) {
      float l1 = 0.f, bm = 0.f;
      for (int i = threadIdx.x; i < d; i += 256) {
        l1 = fmaxf(l1, rowl1[b * d + i]);
        bm = fmaxf(bm, fabsf(bias[b * d + i]));
      }
      l1 = wave_max(l1);
      bm = wave_max(bm);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6] = l1;
        sh[4 + (threadIdx.x >> 6)] = bm;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        blk[2 * b] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        blk[2 * b + 1] = fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]));
      }
    }
    __syncthreads();
    qb = qb * blk[0] + blk[1];
    kb = kb * blk[2] + blk[3];
    vb = vb * blk[4] + blk[5];
  }
  if (threadIdx.x == 0) {
    // ek = half the exponent gap: K 2^ek and Q qscale 2^-ek both sit at sqrt(|q||k|)
    const float qs = qb * qscale;
    int ek = 0;
    if (qs > 0.f && kb > 0.f && qs < 3.0e38f && kb < 3.0e38f) {
      const int eq = (int)((__float_as_uint(qs) >> 23) & 0xff) - 127;
      const int ekk = (int)((__float_as_uint(kb) >> 23) & 0xff) - 127;
      ek = (eq - ekk) >> 1;
      ek = max(-60, min(60, ek));
    }
    const int ev = pow2_exp_for(vb);
    scales[0] = qscale * pow2f(-ek);
    scales[1] = pow2f(ek);
    scales[2] = pow2f(ev);
    scales[3] = pow2f(-ev);
    // the attention output is a convex combination of value rows: |out| <= max |v| < 2^(15 - ev)
    if (out_range) out_range[0] = pow2f(15 - ev);
  }
}

}  // namespace
}  // namespace spr

using namespace spr;

static size_t attn_tp_(int t, int nseg) { return align_up((size_t)t + 16 * (size_t)nseg + KT2, PT); }
// small scratch behind the planes: 3 absmax partial arrays, 3*256 row norms, 4 scales
static constexpr size_t kAttnSmall = 3 * 2048 + 4096 + 256;

extern "C" size_t spr_attn_workspace_bytes(int t, int nseg, int nhead, int head_dim) {
  if (t < 0 || nseg < 0 || nhead < 0 || head_dim < 0) return 0;
  const size_t d = (size_t)nhead * head_dim;
  return 4 * align_up((size_t)(t > 0 ? t : 1) * d * 2, 256) + 2 * align_up(d * attn_tp_(t, nseg) * 2, 256) +
         kAttnSmall;
}

namespace {
struct AttnSmall {
  float *p0, *p1, *p2, *rowl1, *scales;
};
// carves the operand planes and the small scratch out of ws
int carve(void* ws, size_t ws_bytes, int t, int nseg, int d, size_t tp, AttnPlanes& pl, AttnSmall& sm) {
  Workspace w(ws, ws_bytes);
  pl.qh = w.take<_Float16>((size_t)t * d);
  pl.ql = w.take<_Float16>((size_t)t * d);
  pl.kh = w.take<_Float16>((size_t)t * d);
  pl.kl = w.take<_Float16>((size_t)t * d);
  pl.vth = w.take<_Float16>((size_t)d * tp);
  pl.vtl = w.take<_Float16>((size_t)d * tp);
  sm.p0 = w.take<float>(kAmaxParts);
  sm.p1 = w.take<float>(kAmaxParts);
  sm.p2 = w.take<float>(kAmaxParts);
  sm.rowl1 = w.take<float>(1024);
  sm.scales = w.take<float>(4);
  SPR_REQUIRE(sm.scales != nullptr, "attention: workspace carve failed");
  pl.scales = sm.scales;
  return 0;
}

int env_attn_nq() {
  static const int nq = [] { const char* e = getenv("SPR_ATTN_NQ"); return (e != nullptr && e[0] == '2') ? 2 : 1; }();
  return nq;
}
bool env_attn_lazy() {   // lazy softmax reference (k_attn_h3<*, true>) unless SPR_ATTN_LAZY=0 (A/B against the eager form)
  static const bool lazy = [] { const char* e = getenv("SPR_ATTN_LAZY"); return e == nullptr || e[0] != '0'; }();
  return lazy;
}
// does the core launched for `mode` write the per-query log-sum-exp (lse_out)?  Every lazy-softmax form does.
bool attn_core_writes_lse(int mode) { return mode != 0 && (env_attn_nq() == 1 || env_attn_lazy()); }

int launch_core(const AttnPlanes& pl, int t, size_t tp, const int* cu, const int* kv_seg, int nseg,
                int max_len_host, int nhead, float* out, int o_stride, int mode, hipStream_t stream,
                float* lse_out = nullptr, const int* o_tiles = nullptr) {
  ProfScope prof(stream, -1, t);
  // Default since round 4: 32 queries per wave with a 168-register budget, three waves per SIMD (147 VGPRs in split
  // mode): 927 vs 956 us per call at the bench shape, 452 vs 458 us in mode 2.  SPR_ATTN_NQ=2 selects the 64-query /
  // two-wave form.  (Four waves per SIMD -- 118 VGPRs in mode 2, 128 with 42 spilled registers in split mode --
  // measured 452 and 1 171 us: occupancy is not what this kernel waits for)
  const int nq = env_attn_nq();
  // Round 5 default: the sum-checked core k_attn_s (no row-maximum pass; mode 3 = one probability plane).
  // SPR_ATTN_CORE=h3 selects the round-4 kernel for A/B (modes 1 and 2 only); s_setprio 2 around the score MFMAs
  // (-1.5 % in every mode) unless SPR_ATTN_PRIO=0.
  static const bool core_h3 = [] { const char* e = getenv("SPR_ATTN_CORE"); return e != nullptr && e[0] == 'h'; }();
  static const int prio = [] { const char* e = getenv("SPR_ATTN_PRIO"); return e != nullptr ? atoi(e) : 2; }();
  if (mode >= 3 || (nq == 1 && !core_h3)) {
    // mode 4: significance threshold 2^-SPR_ATTN_SIG of the lane's running sum (default 2^-5)
    static const float sig = [] { const char* e = getenv("SPR_ATTN_SIG"); return ldexpf(1.0f, -(e != nullptr ? atoi(e) : 5)); }();
    dim3 grid1(cdiv(max_len_host, QB2 / 2) * nhead * nseg);
#define SPR_ATTN_S(H3_, PLO_, PR_)                                                                                   \
    hipLaunchKernelGGL((k_attn_s<H3_, PLO_, 3, PR_>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth, \
                       pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out, o_tiles, sig)
#ifdef SPR_ATTN_ABLATE
    static const int abl = [] { const char* e = getenv("SPR_ATTN_ABL"); return e != nullptr ? atoi(e) : 0; }();
#define SPR_ATTN_SA(A_)                                                                                               \
    hipLaunchKernelGGL((k_attn_s<true, true, 3, 2, A_>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth, \
                       pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out, o_tiles, sig)
    if (abl != 0 && mode == 1) {
      switch (abl) {
        case 1: SPR_ATTN_SA(1); break;
        case 4: SPR_ATTN_SA(4); break;
        case 8: SPR_ATTN_SA(8); break;
        case 16: SPR_ATTN_SA(16); break;
        case 32: SPR_ATTN_SA(32); break;
        case 64: SPR_ATTN_SA(64); break;
        default: SPR_ATTN_SA(7); break;
      }
      SPR_LAUNCH_CHECK();
      return 0;
    }
#undef SPR_ATTN_SA
#endif
    static const int pf = [] { const char* e = getenv("SPR_ATTN_PF"); return e != nullptr ? atoi(e) : 0; }();
    if (pf > 0 && (mode == 1 || mode == 3)) {
      if (mode == 1)
        hipLaunchKernelGGL((k_attn_s<true, true, 3, 2, 0, false, 1>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl,
                           pl.vth, pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out, o_tiles, sig);
      else
        hipLaunchKernelGGL((k_attn_s<true, false, 3, 2, 0, false, 1>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl,
                           pl.vth, pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out, o_tiles, sig);
      SPR_LAUNCH_CHECK();
      return 0;
    }
    if (mode == 4) {
      static const int awps = [] { const char* e = getenv("SPR_ATTN_ADAPT_WPS"); return e != nullptr ? atoi(e) : 4; }();
      if (awps == 3)
        hipLaunchKernelGGL((k_attn_s<true, true, 3, 2, 0, true>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                           pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out, o_tiles, sig);
      else
      hipLaunchKernelGGL((k_attn_s<true, true, 4, 2, 0, true>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                         pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out, o_tiles, sig);
    } else if (prio > 0) {
      if (mode == 2) SPR_ATTN_S(false, false, 2);
      else if (mode == 3) SPR_ATTN_S(true, false, 2);
      else SPR_ATTN_S(true, true, 2);
    } else {
      if (mode == 2) SPR_ATTN_S(false, false, 0);
      else if (mode == 3) SPR_ATTN_S(true, false, 0);
      else SPR_ATTN_S(true, true, 0);
    }
#undef SPR_ATTN_S
    SPR_LAUNCH_CHECK();
    return 0;
  }
  SPR_REQUIRE(o_tiles == nullptr, "attention core: tiled output needs the k_attn_s core (attn_core_tiled_ok)");
  if (nq == 1) {
    dim3 grid1(cdiv(max_len_host, QB2 / 2) * nhead * nseg);
    // experiment switch (profiles/r04_attn_counters.txt): the same kernel compiled for FOUR waves per SIMD
    static const bool wps4 = [] { const char* e = getenv("SPR_ATTN_WPS"); return e != nullptr && e[0] == '4'; }();
    if (wps4) {
      if (mode == 2)
        hipLaunchKernelGGL((k_attn_h3<false, true, 1, 4>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                           pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
      else
        hipLaunchKernelGGL((k_attn_h3<true, true, 1, 4>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                           pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
      SPR_LAUNCH_CHECK();
      return 0;
    }
    static const int pipe = [] { const char* e = getenv("SPR_ATTN_PIPE"); return e != nullptr ? atoi(e) : 0; }();
    if (pipe == 1) {        // experiment: the two key halves of a tile staggered inside the wave (tile_pipe)
      if (mode == 2)
        hipLaunchKernelGGL((k_attn_h3<false, true, 1, 3, true>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl,
                           pl.vth, pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
      else
        hipLaunchKernelGGL((k_attn_h3<true, true, 1, 3, true>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl,
                           pl.vth, pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
      SPR_LAUNCH_CHECK();
      return 0;
    }
    if (pipe == 2) {        // the same at two waves per SIMD (256 registers)
      if (mode == 2)
        hipLaunchKernelGGL((k_attn_h3<false, true, 1, 2, true>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl,
                           pl.vth, pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
      else
        hipLaunchKernelGGL((k_attn_h3<true, true, 1, 2, true>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl,
                           pl.vth, pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
      SPR_LAUNCH_CHECK();
      return 0;
    }
    if (mode == 2)
      hipLaunchKernelGGL((k_attn_h3<false, true, 1, 3>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                         pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
    else
      hipLaunchKernelGGL((k_attn_h3<true, true, 1, 3>), grid1, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                         pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(cdiv(max_len_host, QB2) * nhead * nseg);
  // lazy softmax reference (k_attn_h3<*, true>) unless SPR_ATTN_LAZY=0 (A/B timing against the eager form)
  const bool lazy = env_attn_lazy();
  if (lazy) {
    if (mode == 2)
      hipLaunchKernelGGL((k_attn_h3<false, true>), grid, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                         pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
    else
      hipLaunchKernelGGL((k_attn_h3<true, true>), grid, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth,
                         pl.vtl, t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  if (mode == 2)
    hipLaunchKernelGGL(k_attn_h3<false>, grid, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth, pl.vtl,
                       t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
  else
    hipLaunchKernelGGL(k_attn_h3<true>, grid, dim3(256), 0, stream, pl.qh, pl.ql, pl.kh, pl.kl, pl.vth, pl.vtl,
                       t, (int)tp, cu, kv_seg, nseg, nhead, pl.scales, out, o_stride, lse_out);
  SPR_LAUNCH_CHECK();
  return 0;
}
}  // namespace

static int attn_varlen_fwd_impl(const float* q, int q_stride, const float* k, int k_stride,
                               const float* v, int v_stride, const int* cu,
                               const int* kv_seg, int t, int nseg, int max_len_host, int nhead,
                               int head_dim, float scale, float* out, int o_stride, float* lse_out, void* ws,
                               size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(head_dim == HD, "attention: head_dim must be %d (got %d)", HD, head_dim);
  SPR_REQUIRE(t >= 1 && nseg >= 1 && nhead >= 1 && max_len_host >= 1, "attention: bad sizes");
  SPR_REQUIRE(q_stride % 4 == 0 && k_stride % 4 == 0 && v_stride % 4 == 0 && o_stride % 4 == 0,
              "attention: row strides must be multiples of 4 floats");
  SPR_REQUIRE((long)cdiv(max_len_host, QB) * nhead * nseg < (1l << 31), "attention: grid too large");
  const int mode = spr::g_attn_mode.load(std::memory_order_relaxed);
  if (mode == 0) {
    dim3 grid(cdiv(max_len_host, QB) * nhead * nseg);
    hipLaunchKernelGGL(k_attn, grid, dim3(256), 0, stream, q, q_stride, k, k_stride, v, v_stride, cu,
                       kv_seg, nseg, nhead, scale, out, o_stride);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  const int d_model = nhead * head_dim;
  const size_t tp = attn_tp_(t, nseg);
  SPR_REQUIRE(tp < (1ul << 31), "attention: too many tokens");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_attn_workspace_bytes(t, nseg, nhead, head_dim),
              "attention: workspace too small (%zu bytes given)", ws_bytes);
  AttnPlanes pl{};
  AttnSmall sm{};
  if (int rc = carve(ws, ws_bytes, t, nseg, d_model, tp, pl, sm)) return rc;
  if (int rc = launch_absmax2(q, t, d_model, q_stride, sm.p0, k, t, d_model, k_stride, sm.p1, stream)) return rc;
  if (int rc = launch_absmax(v, t, d_model, v_stride, sm.p2, stream)) return rc;
  hipLaunchKernelGGL(k_plane_scales, dim3(1), dim3(256), 0, stream, sm.p0, sm.p1, sm.p2, (const float*)nullptr,
                     (const float*)nullptr, d_model, scale * 1.4426950408889634f, sm.scales, kAmaxParts, kAmaxParts,
                     kAmaxParts, (float*)nullptr);
  hipLaunchKernelGGL(k_attn_pack, dim3((unsigned)(tp / PT)), dim3(256), 0, stream, q, q_stride, k, k_stride,
                     v, v_stride, cu, nseg, d_model, t, (int)tp, sm.scales, pl.qh, pl.ql, pl.kh, pl.kl,
                     pl.vth, pl.vtl);
  SPR_LAUNCH_CHECK();
  return launch_core(pl, t, tp, cu, kv_seg, nseg, max_len_host, nhead, out, o_stride, mode, stream, lse_out);
}

extern "C" int spr_attn_varlen_fwd(const float* q, int q_stride, const float* k, int k_stride,
                                   const float* v, int v_stride, const int* cu,
                                   const int* kv_seg, int t, int nseg, int max_len_host, int nhead,
                                   int head_dim, float scale, float* out, int o_stride, void* ws,
                                   size_t ws_bytes, void* stream_) {
  return attn_varlen_fwd_impl(q, q_stride, k, k_stride, v, v_stride, cu, kv_seg, t, nseg, max_len_host, nhead, head_dim,
                              scale, out, o_stride, nullptr, ws, ws_bytes, stream_);
}

// The same, additionally handing out lse [t, nhead] = log2 sum_j 2^(log2(e) q_i.k_j scale) per query and head for
// spr_attn_varlen_bwd_lse (the backward then skips its own pass over the keys).  *lse_written = 0 when the
// configured core does not produce it (exact-f32 mode, the eager-softmax experiment): lse is then untouched.
extern "C" int spr_attn_varlen_fwd_lse(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                                       int v_stride, const int* cu, const int* kv_seg, int t, int nseg,
                                       int max_len_host, int nhead, int head_dim, float scale, float* out, int o_stride,
                                       float* lse, int* lse_written, void* ws, size_t ws_bytes, void* stream_) {
  SPR_REQUIRE(lse != nullptr && lse_written != nullptr, "attention: lse / lse_written must not be null");
  const bool can = attn_core_writes_lse(spr::g_attn_mode.load(std::memory_order_relaxed));
  *lse_written = can ? 1 : 0;
  return attn_varlen_fwd_impl(q, q_stride, k, k_stride, v, v_stride, cu, kv_seg, t, nseg, max_len_host, nhead, head_dim,
                              scale, out, o_stride, can ? lse : nullptr, ws, ws_bytes, stream_);
}

extern "C" size_t spr_attn_inproj_workspace_bytes(int t, int nseg, int nhead, int head_dim) {
  if (t < 0 || nseg < 0 || nhead < 0 || head_dim < 0) return 0;
  // operand planes + (exact mode only) the fp32 [t, 3 d] projection
  return spr_attn_workspace_bytes(t, nseg, nhead, head_dim) +
         align_up((size_t)(t > 0 ? t : 1) * 3 * nhead * head_dim * sizeof(float), 256);
}

extern "C" int spr_attn_inproj_varlen_fwd(const float* x_qk, const float* x_v, int t, const float* w_in,
                                          const float* b_in, const int* cu, const int* kv_seg, int nseg,
                                          int max_len_host, int nhead, int head_dim, float scale, float* out,
                                          int o_stride, void* ws, size_t ws_bytes, void* stream_) {
  return spr_attn_inproj_varlen_fwd_r(x_qk, x_v, t, w_in, b_in, cu, kv_seg, nseg, max_len_host, nhead, head_dim, scale,
                                      out, o_stride, nullptr, 0, nullptr, 0, nullptr, nullptr, ws, ws_bytes, stream_);
}

// Weight-side inputs of the fused in-projection, measured once per weight version:
// out[0 .. spr_range_parts()) = max |w| partials, then 3 d row L1 norms of w_in [3d, d].
extern "C" int spr_attn_inproj_prepare(const float* w_in, int d, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(w_in != nullptr && out != nullptr && d >= 1 && 3 * d <= 1024, "inproj_prepare: bad arguments");
  if (int rc = launch_absmax(w_in, 3 * d, d, d, out, stream)) return rc;
  hipLaunchKernelGGL(k_row_l1, dim3(cdiv((long)3 * d * 64, 256)), dim3(256), 0, stream, w_in, 3 * d, d, out + kAmaxParts);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_attn_inproj_varlen_fwd_r(const float* x_qk, const float* x_v, int t, const float* w_in,
                                            const float* b_in, const int* cu, const int* kv_seg, int nseg,
                                            int max_len_host, int nhead, int head_dim, float scale, float* out,
                                            int o_stride, const float* xqk_range, int xqk_range_n,
                                            const float* xv_range, int xv_range_n, float* out_range,
                                            const float* w_prep, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(head_dim == HD, "attention: head_dim must be %d (got %d)", HD, head_dim);
  SPR_REQUIRE(nhead * head_dim == 256, "attention in-projection: d_model must be 256 (got %d)", nhead * head_dim);
  SPR_REQUIRE(t >= 1 && nseg >= 1 && max_len_host >= 1, "attention: bad sizes");
  SPR_REQUIRE(x_qk && x_v && w_in && b_in, "attention in-projection: null operand");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_attn_inproj_workspace_bytes(t, nseg, nhead, head_dim),
              "attention in-projection: workspace too small (%zu bytes given)", ws_bytes);
  const int d = nhead * head_dim;
  const size_t planes_bytes = spr_attn_workspace_bytes(t, nseg, nhead, head_dim);
  const int mode = spr::g_attn_mode.load(std::memory_order_relaxed);
  const size_t tp = attn_tp_(t, nseg);
  SPR_REQUIRE(tp < (1ul << 31), "attention: too many tokens");
  AttnPlanes pl{};
  AttnSmall sm{};
  if (int rc = carve(ws, planes_bytes, t, nseg, d, tp, pl, sm)) return rc;
  const bool gemm_split = gemm_mode() == 1;
  if (mode == 0 || t < 256 || !gemm_split) {
    if (out_range != nullptr) {   // this route publishes no range: an infinite bound would be wrong, so measure later
      SPR_REQUIRE(false, "attention in-projection: out_range is only available on the fused split-fp16 path "
                         "(attn mode %d, gemm split %d, t=%d)", mode, (int)gemm_split, t);
    }
    // exact-f32 mode (and tiny inputs): plain projection into the workspace, then the
    // unfused core -- same results as spr_linear + spr_attn_varlen_fwd
    float* qkv = (float*)((char*)ws + planes_bytes);
    const float *xp = nullptr, *xvp = nullptr, *wp = nullptr;
    if (gemm_split) {   // operand ranges for the split-fp16 GEMM (sm.* is rewritten by the core afterwards)
      if (int rc = launch_absmax(x_qk, t, d, d, sm.p0, stream)) return rc;
      if (int rc = launch_absmax(w_in, 3 * d, d, d, sm.p1, stream)) return rc;
      xp = xvp = sm.p0;
      wp = sm.p1;
      if (x_v != x_qk) {
        if (int rc = launch_absmax(x_v, t, d, d, sm.p2, stream)) return rc;
        xvp = sm.p2;
      }
    }
    if (x_v == x_qk) {
      if (int rc = launch_linear_ranged(x_qk, t, d, w_in, 3 * d, b_in, qkv, xp, wp, stream)) return rc;
    } else {
      // two projections, written side by side: [q | k] from x_qk, [v] from x_v
      float* qk = qkv;
      float* v = qkv + (size_t)t * 2 * d;
      if (int rc = launch_linear_ranged(x_qk, t, d, w_in, 2 * d, b_in, qk, xp, wp, stream)) return rc;
      if (int rc = launch_linear_ranged(x_v, t, d, w_in + (size_t)2 * d * d, d, b_in + 2 * d, v, xvp, wp, stream))
        return rc;
      return spr_attn_varlen_fwd(qk, 2 * d, qk + d, 2 * d, v, d, cu, kv_seg, t, nseg, max_len_host, nhead,
                                 head_dim, scale, out, o_stride, ws, planes_bytes, stream_);
    }
    return spr_attn_varlen_fwd(qkv, 3 * d, qkv + d, 3 * d, qkv + 2 * d, 3 * d, cu, kv_seg, t, nseg,
                               max_len_host, nhead, head_dim, scale, out, o_stride, ws, planes_bytes, stream_);
  }
  SPR_REQUIRE((long)cdiv(max_len_host, QB2 / 2) * nhead * nseg < (1l << 31), "attention: grid too large");
  pl.cu = cu;
  pl.nseg = nseg;
  pl.t_total = t;
  pl.tp = (int)tp;
  // operand ranges: max|x| (inputs), max|w| (the projection's own operand scale) and the plane
  // multipliers from the bounds max|x| * max row-L1(W block) + max|bias block|
  // input ranges: published by the producer (LayerNorm) or measured here
  // weight side (max |w| partials + row L1 norms): measured here, or once per weight version by
  // the caller (spr_attn_inproj_prepare -> w_prep)
  const float* wparts = w_prep != nullptr ? w_prep : sm.p1;
  const float* rowl1 = w_prep != nullptr ? w_prep + kAmaxParts : sm.rowl1;
  const float* xqp = sm.p0;
  int n_xq = kAmaxParts;
  if (xqk_range != nullptr) {
    SPR_REQUIRE(xqk_range_n >= 1, "attention in-projection: xqk_range needs a count");
    xqp = xqk_range;
    n_xq = xqk_range_n;
    if (w_prep == nullptr)
      if (int rc = launch_absmax(w_in, 3 * d, d, d, sm.p1, stream)) return rc;
  } else if (w_prep == nullptr) {
    if (int rc = launch_absmax2(x_qk, t, d, d, sm.p0, w_in, 3 * d, d, d, sm.p1, stream)) return rc;
  } else {
    if (int rc = launch_absmax(x_qk, t, d, d, sm.p0, stream)) return rc;
  }
  const float* xvp = xqp;
  int n_xv = n_xq;
  if (x_v != x_qk) {
    if (xv_range != nullptr) {
      SPR_REQUIRE(xv_range_n >= 1, "attention in-projection: xv_range needs a count");
      xvp = xv_range;
      n_xv = xv_range_n;
    } else {
      if (int rc = launch_absmax(x_v, t, d, d, sm.p2, stream)) return rc;
      xvp = sm.p2;
      n_xv = kAmaxParts;
    }
  }
  if (w_prep == nullptr)
    hipLaunchKernelGGL(k_row_l1, dim3(cdiv((long)3 * d * 64, 256)), dim3(256), 0, stream, w_in, 3 * d, d, sm.rowl1);
  hipLaunchKernelGGL(k_plane_scales, dim3(1), dim3(256), 0, stream, xqp, xqp, xvp, rowl1, b_in, d,
                     scale * 1.4426950408889634f, sm.scales, n_xq, n_xq, n_xv, out_range);
  hipLaunchKernelGGL(k_attn_zero_gaps, dim3(d), dim3(256), 0, stream, cu, nseg, (int)tp, pl.vth, pl.vtl);
  if (x_v == x_qk) {
    if (int rc = launch_inproj_planes(x_qk, t, d, w_in, 3 * d, b_in, 0, pl, xqp, n_xq, wparts, stream)) return rc;
  } else {
    if (int rc = launch_inproj_planes(x_qk, t, d, w_in, 2 * d, b_in, 0, pl, xqp, n_xq, wparts, stream)) return rc;
    if (int rc = launch_inproj_planes(x_v, t, d, w_in + (size_t)2 * d * d, d, b_in + 2 * d, 2 * d, pl, xvp, n_xv, wparts,
                                      stream))
      return rc;
  }
  return launch_core(pl, t, tp, cu, kv_seg, nseg, max_len_host, nhead, out, o_stride, mode, stream);
}

// ---- entry points for the fused cross-encoder chains (xenc.hip) --------------------------------
size_t spr::attn_tp(int t, int nseg) { return ::attn_tp_(t, nseg); }
int spr::attn_mode() { return spr::g_attn_mode.load(std::memory_order_relaxed); }
int spr::attn_carve_planes(void* ws, size_t ws_bytes, int t, int nseg, int d, AttnPlanes& pl) {
  AttnSmall sm{};
  return carve(ws, ws_bytes, t, nseg, d, ::attn_tp_(t, nseg), pl, sm);
}
int spr::attn_zero_gaps(const AttnPlanes& pl, int d, hipStream_t stream) {
  hipLaunchKernelGGL(k_attn_zero_gaps, dim3(d), dim3(256), 0, stream, pl.cu, pl.nseg, pl.tp, pl.vth, pl.vtl);
  SPR_LAUNCH_CHECK();
  return 0;
}
// does launch_core() pick k_attn_s (the only core that can write the chains' tiled output) for `mode`?
bool spr::attn_core_tiled_ok(int mode) {
  static const bool core_h3 = [] { const char* e = getenv("SPR_ATTN_CORE"); return e != nullptr && e[0] == 'h'; }();
  return mode >= 3 || ((mode == 1 || mode == 2) && env_attn_nq() == 1 && !core_h3);
}
int spr::attn_core_on_planes(const AttnPlanes& pl, const int* kv_seg, int max_len_host, int nhead, float* out,
                             int o_stride, int mode, hipStream_t stream, const int* o_tiles) {
  SPR_REQUIRE(mode >= 1 && mode <= 4, "attention core on planes: mode must be 1 .. 4 (got %d)", mode);
  SPR_REQUIRE((long)cdiv(max_len_host, QB2 / 2) * nhead * pl.nseg < (1l << 31), "attention: grid too large");
  SPR_REQUIRE(o_tiles == nullptr || (nhead == 8 && attn_core_tiled_ok(mode)), "attention core: tiled output unavailable");
  return launch_core(pl, pl.t_total, (size_t)pl.tp, pl.cu, kv_seg, pl.nseg, max_len_host, nhead, out, o_stride, mode,
                     stream, nullptr, o_tiles);
}

extern "C" int spr_set_attn_mode(int mode) {
  SPR_REQUIRE(mode >= 0 && mode <= 4,
              "attention mode must be 0 (exact f32 MFMA), 1 (split-fp16), 2 (single-pass fp16), 3 (split-fp16 "
              "scores, one probability plane) or 4 (as 1, the lo plane of the probabilities only on significant tiles)");
  spr::g_attn_mode.store(mode);
  return 0;
}
