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
#include "spr_common.h"

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
// Split-fp16 variant ("h3"): same structure, fp32-level accuracy, ~5x less
// matrix-core time.  Every operand x is carried as hi = fp16(x) and
// lo = fp16(x - hi); a.b ~= ah.bh + ah.bl + al.bh in ONE fp32 accumulator
// (the matrix cores honour fp16 subnormals, scripts/abl/denorm.hip, so lo needs
// no rescaling; its absolute resolution is 2^-24).
//   S^T = K Q^T : 2 k-steps x 3 v_mfma_f32_32x32x16_f16  (head_dim 32)
//   O^T = V^T P^T: the f32 probabilities in the S^T accumulator are converted
//     in place -- registers 8s..8s+7 of a lane are exactly the B fragment of
//     k-step s, with key order kappa(s,h,j) = 16s + 8(j>>2) + 4h + (j&3); the
//     V^T A-fragment is read with the same order from a TRANSPOSED fp16 V tile
//     (two 8-byte reads).  P is scaled by 2^10 before the split so that small
//     probabilities stay in fp16's normal range (undone in the final 1/l).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int KH = 40;   // K tile row stride in halves (80 B: conflict-free ds_read_b128)
constexpr int VH = 36;   // V^T tile row stride in halves (72 B: conflict-free ds_read_b64)

__device__ __forceinline__ void split_h(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

__global__ __launch_bounds__(256) void k_attn_h3(
    const float* __restrict__ q, int q_stride, const float* __restrict__ k, int k_stride,
    const float* __restrict__ v, int v_stride, const int* __restrict__ cu,
    const int* __restrict__ kv_seg, int nseg, int nhead, float scale, float* __restrict__ out,
    int o_stride) {
  __shared__ __align__(16) _Float16 Kh[2][KT * KH], Kl[2][KT * KH];
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
  const int q0 = qt * QB;
  if (q0 >= qlen) return;
  const int ks = kv_seg[seg];
  const int kbeg = cu[ks], klen = cu[ks + 1] - kbeg;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int qi = q0 + wave * QW + l31;
  const bool qok = qi < qlen;
  const int hoff = head * HD;

  // Q^T B-fragments: lane (query l31, half lh), k-step s: d = 16 s + 8 lh + j
  h16x8 qh[2], ql[2];
  {
    const float sc = scale * 1.4426950408889634f;   // base-2 softmax
    const float* qp = q + (size_t)(qbeg + (qok ? qi : 0)) * q_stride + hoff + 8 * lh;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = qok ? qp[16 * s + j] * sc : 0.f;
        _Float16 a, b;
        split_h(x, a, b);
        qh[s][j] = a;
        ql[s][j] = b;
      }
  }

  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int sr = tid >> 3, sc4 = (tid & 7) * 4;   // staging role: key row, 4 dims
  f32x4 kreg, vreg;
  auto fetch = [&](int kt) {
    const int r = min(kt + sr, klen - 1);
    const size_t row = (size_t)(kbeg + r);
    const float* kp_ = k + row * k_stride + hoff + sc4;
    const float* vp_ = v + row * v_stride + hoff + sc4;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kreg) : "v"(kp_));
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vreg) : "v"(vp_));
  };
  auto stash = [&](int kt, int buf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const bool in = kt + sr < klen;
    h16x4 kh4, kl4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      _Float16 a, b;
      split_h(in ? kreg[e] : 0.f, a, b);
      kh4[e] = a;
      kl4[e] = b;
      split_h(in ? vreg[e] : 0.f, a, b);
      Vth[buf][(sc4 + e) * VH + sr] = a;     // transposed: [d][key]
      Vtl[buf][(sc4 + e) * VH + sr] = b;
    }
    *reinterpret_cast<h16x4*>(Kh[buf] + sr * KH + sc4) = kh4;
    *reinterpret_cast<h16x4*>(Kl[buf] + sr * KH + sc4) = kl4;
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

    // ---- S^T = K Q^T (rows = keys, cols = queries) ----
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
    h16x8 kfh[2], kfl[2], vfh[2], vfl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      kfh[s] = *reinterpret_cast<const h16x8*>(Kh[buf] + l31 * KH + 16 * s + 8 * lh);
      kfl[s] = *reinterpret_cast<const h16x8*>(Kl[buf] + l31 * KH + 16 * s + 8 * lh);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], ql[s], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfl[s], qh[s], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kfh[s], qh[s], sacc, 0, 0, 0);
    }
    // V^T A-fragments (issued early: their LDS latency hides under the softmax)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const _Float16* ph_ = Vth[buf] + l31 * VH + 16 * s + 4 * lh;
      const _Float16* pl_ = Vtl[buf] + l31 * VH + 16 * s + 4 * lh;
      const h16x4 a0 = *reinterpret_cast<const h16x4*>(ph_);
      const h16x4 a1 = *reinterpret_cast<const h16x4*>(ph_ + 8);
      const h16x4 b0 = *reinterpret_cast<const h16x4*>(pl_);
      const h16x4 b1 = *reinterpret_cast<const h16x4*>(pl_ + 8);
      vfh[s] = (h16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      vfl[s] = (h16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }

    // ---- online softmax (lane = query; reg r <-> key (r&3) + 8 (r>>2) + 4 lh) ----
    if (kt + KT > klen) {             // wave-uniform: only the last tile has a tail
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (j >= klen) sacc[r] = -INFINITY;
      }
    }
    float mx = sacc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    // the running maximum rarely moves after the first tiles: skip the rescale
    // of the 16 output registers unless some lane of the wave needs it
    if (__ballot(m_new != m_run) != 0ull) {
      const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= corr;
      m_run = m_new;
    }
    // p' = 2^10 * exp2(s - m): the 2^10 keeps small probabilities inside fp16's
    // normal range; l_run accumulates the same scaled values, so it cancels in
    // the final 1/l.  hi by packed round-toward-zero conversion, lo = p' - hi.
    const float mshift = m_run - 10.0f;
    float psum = 0.f;
    unsigned int ph_u[8], pl_u[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __builtin_amdgcn_exp2f(sacc[r] - mshift);
      const float p1 = __builtin_amdgcn_exp2f(sacc[r + 1] - mshift);
      psum += p0 + p1;
      const h16x2 hi = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));
      const h16x2 lo = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0 - (float)hi[0], p1 - (float)hi[1]));
      ph_u[r >> 1] = __builtin_bit_cast(unsigned int, hi);
      pl_u[r >> 1] = __builtin_bit_cast(unsigned int, lo);
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run += psum;
    h16x8 pbh[2], pbl[2];   // registers 8s..8s+7 = B fragment of k-step s
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const u32x4 a = {ph_u[4 * s2], ph_u[4 * s2 + 1], ph_u[4 * s2 + 2], ph_u[4 * s2 + 3]};
      const u32x4 b = {pl_u[4 * s2], pl_u[4 * s2 + 1], pl_u[4 * s2 + 2], pl_u[4 * s2 + 3]};
      pbh[s2] = __builtin_bit_cast(h16x8, a);
      pbl[s2] = __builtin_bit_cast(h16x8, b);
    }

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[s], pbl[s], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl[s], pbh[s], o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh[s], pbh[s], o, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) stash(kt + KT, buf ^ 1);
    __syncthreads();
  }

  if (qok) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
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

static int g_attn_mode = 1;   // 1 = split-fp16 (default), 0 = exact f32 MFMA

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" int spr_attn_varlen_fwd(const float* q, int q_stride, const float* k, int k_stride,
                                   const float* v, int v_stride, const int* cu,
                                   const int* kv_seg, int nseg, int max_len_host, int nhead,
                                   int head_dim, float scale, float* out, int o_stride,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(head_dim == HD, "attention: head_dim must be %d (got %d)", HD, head_dim);
  SPR_REQUIRE(nseg >= 1 && nhead >= 1 && max_len_host >= 1, "attention: bad sizes");
  SPR_REQUIRE(q_stride % 4 == 0 && k_stride % 4 == 0 && v_stride % 4 == 0 && o_stride % 4 == 0,
              "attention: row strides must be multiples of 4 floats");
  SPR_REQUIRE((long)cdiv(max_len_host, QB) * nhead * nseg < (1l << 31), "attention: grid too large");
  dim3 grid(cdiv(max_len_host, QB) * nhead * nseg);
  if (spr::g_attn_mode == 1)
    hipLaunchKernelGGL(k_attn_h3, grid, dim3(256), 0, stream, q, q_stride, k, k_stride, v, v_stride, cu,
                       kv_seg, nseg, nhead, scale, out, o_stride);
  else
    hipLaunchKernelGGL(k_attn, grid, dim3(256), 0, stream, q, q_stride, k, k_stride, v, v_stride, cu,
                       kv_seg, nseg, nhead, scale, out, o_stride);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_set_attn_mode(int mode) {
  SPR_REQUIRE(mode == 0 || mode == 1, "attention mode must be 0 (exact f32 MFMA) or 1 (split-fp16)");
  spr::g_attn_mode = mode;
  return 0;
}
