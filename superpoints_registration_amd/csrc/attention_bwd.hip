// 8f-1 -- backward of the varlen multi-head attention core (spr_attn_varlen_fwd; the graph torch autograd
// builds for F.multi_head_attention_forward at transformer/transformers.py:198-227), flash style: nothing of
// size Lq x Lk is ever written.  Round 3 replaces the materialised form (S, P, dP, dS per (segment, head) as
// batches of spr_bgemm + two softmax passes: 56 launches and ~2 GB of traffic per attention call).
//
//   D[t, h]  = sum_d dO[t, h, d] O[t, h, d]                                           k_attn_bwd_rowdot
//   L[t, h]  = log2 sum_j exp2(c s_tj),  c = log2(e) / sqrt(d)       (sweep 1 of)      k_attn_bwd_dq
//   dQ       = scale . dS K,    dS = P o (dP - D),  P = exp2(c S - L), dP = dO V^T    (sweep 2 of) k_attn_bwd_dq
//   dV = P^T dO,  dK = scale . dS^T Q                                                 k_attn_bwd_dkv
//
// Arithmetic: attention mode 0 (spr_set_attn_mode): exact f32 MFMA (v_mfma_f32_32x32x2_f32) like spr_bgemm; mode 1 / 2
// (default): the split-fp16 form further down.  fp32 softmax with v_exp_f32 in both; fixed summation order ->
// bitwise reproducible gradients.
//
// Layout trick (no LDS round trip for the probabilities): the 32x32 C tile holds column n = lane % 32 and the
// rows 8 (r / 4) + 4 (lane / 32) + r % 4 in registers r = 0..15.  A C tile whose ROWS are the contraction
// index of the next product can be fed to that product as its B operand, register by register, when the A
// operand walks the contraction index in the same order -- k-step r contracts the row pair
// (row(r, 0), row(r, 1)).  Hence k_attn_bwd_dq works on transposed tiles S^T, dP^T (rows = keys: dQ^T = K^T dS^T
// contracts over keys) and k_attn_bwd_dkv on S, dP (rows = queries: dV^T = dO^T P and dK^T = Q^T dS contract
// over queries).
#include "attn_planes.h"
#include "spr_common.h"

namespace spr {
namespace {

constexpr int BHD = 32;    // head dimension
constexpr int BT = 64;     // tile: 64 queries x 64 keys per workgroup iteration (one 32x32 sub-tile per wave)
constexpr int LS = 36;     // LDS row stride in floats: 16-byte aligned rows, conflict-free column reads
constexpr float kLog2e = 1.4426950408889634f;

struct AttnBwdArgs {
  const float *q, *k, *v, *out, *dout;
  int qs, ks, vs, os, dos;       // row strides (floats)
  const int* cu;
  const int* kv_seg;             // key segment of a query segment
  const int* q_seg;              // query segment of a key segment (inverse permutation)
  int nseg, nhead;
  float scale;
  float *lse, *dsum;             // [T, nhead]
  float *dq, *dk, *dv;           // [T, nhead * 32] contiguous
};

__device__ __forceinline__ int crow(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }

__global__ __launch_bounds__(256) void k_attn_bwd_rowdot(const float* __restrict__ out, int os, const float* __restrict__ dout,
                                                         int dos, int t_total, int nhead, float* __restrict__ dsum) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)t_total * nhead) return;
  const int t = (int)(i / nhead), h = (int)(i % nhead);
  const float4* a = reinterpret_cast<const float4*>(out + (size_t)t * os + h * BHD);
  const float4* b = reinterpret_cast<const float4*>(dout + (size_t)t * dos + h * BHD);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < BHD / 4; ++j) {
    const float4 x = a[j], y = b[j];
    s += x.x * y.x;
    s += x.y * y.y;
    s += x.z * y.z;
    s += x.w * y.w;
  }
  dsum[i] = s;
}

// stages a [64 x 32] row tile (rows row0 .. row0 + 63 of a segment with `len` rows; rows past the end are
// zero) of a [T, stride] tensor (head slice at column hoff): two float4 per thread
struct TileRegs {
  float4 a, b;
};
__device__ __forceinline__ TileRegs tile_fetch(const float* __restrict__ x, int stride, int seg_beg, int len, int row0,
                                               int hoff) {
  TileRegs t;
  const int e0 = threadIdx.x, e1 = threadIdx.x + 256;
  const int r0 = row0 + (e0 >> 3), r1 = row0 + (e1 >> 3);
  t.a = make_float4(0.f, 0.f, 0.f, 0.f);
  t.b = t.a;
  if (r0 < len) t.a = reinterpret_cast<const float4*>(x + (size_t)(seg_beg + r0) * stride + hoff)[e0 & 7];
  if (r1 < len) t.b = reinterpret_cast<const float4*>(x + (size_t)(seg_beg + r1) * stride + hoff)[e1 & 7];
  return t;
}
__device__ __forceinline__ void tile_store(float* lds, const TileRegs& t) {
  const int e0 = threadIdx.x, e1 = threadIdx.x + 256;
  *reinterpret_cast<float4*>(lds + (e0 >> 3) * LS + 4 * (e0 & 7)) = t.a;
  *reinterpret_cast<float4*>(lds + (e1 >> 3) * LS + 4 * (e1 & 7)) = t.b;
}

// C = sum_t A_t B_t with A rows read from an LDS tile: lane (m = l31, h) takes x[row0 + l31][16 h + t]
__device__ __forceinline__ f32x16 mm_rows(const float* tile, int row0, int l31, int h, const float (&breg)[16]) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  const float* p = tile + (row0 + l31) * LS + 16 * h;
  float a[16];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(p + 4 * j);
    a[4 * j] = v.x;
    a[4 * j + 1] = v.y;
    a[4 * j + 2] = v.z;
    a[4 * j + 3] = v.w;
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], breg[t], c, 0, 0, 0);
  return c;
}
// acc += X^T . B where B is a C-layout tile (contraction over its rows): A[m = l31][k = (r, h)] =
// tile[row0 + crow(r, h)][l31]
__device__ __forceinline__ void mm_cols_acc(f32x16& acc, const float* tile, int row0, int l31, int h, const f32x16& b) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float a = tile[(row0 + crow(r, h)) * LS + l31];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[r], acc, 0, 0, 0);
  }
}

// grid (query tile, head, query segment)
__global__ __launch_bounds__(256) void k_attn_bwd_dq(const AttnBwdArgs a) {
  __shared__ __align__(16) float Ks[2][BT * LS], Vs[2][BT * LS];
  __shared__ float red_m[2][BT], red_l[2][BT];
  __shared__ float red_acc[2][16][64];
  const int head = blockIdx.y, seg = blockIdx.z;
  const int qbeg = a.cu[seg], qlen = a.cu[seg + 1] - qbeg;
  const int q0 = blockIdx.x * BT;
  if (q0 >= qlen) return;
  const int ksg = a.kv_seg[seg];
  const int kbeg = a.cu[ksg], klen = a.cu[ksg + 1] - kbeg;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int qb = wave & 1, kb = wave >> 1;
  const int hoff = head * BHD;
  const float c = a.scale * kLog2e;

  const int qi = q0 + 32 * qb + l31;
  const bool qvalid = qi < qlen;
  const int qic = qvalid ? qi : qlen - 1;
  float qreg[16], doreg[16];
  {
    const float4* pq = reinterpret_cast<const float4*>(a.q + (size_t)(qbeg + qic) * a.qs + hoff + 16 * h);
    const float4* pd = reinterpret_cast<const float4*>(a.dout + (size_t)(qbeg + qic) * a.dos + hoff + 16 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 x = pq[j], y = pd[j];
      qreg[4 * j] = x.x; qreg[4 * j + 1] = x.y; qreg[4 * j + 2] = x.z; qreg[4 * j + 3] = x.w;
      doreg[4 * j] = y.x; doreg[4 * j + 1] = y.y; doreg[4 * j + 2] = y.z; doreg[4 * j + 3] = y.w;
    }
  }
  const float dq_row = a.dsum[(size_t)(qbeg + qic) * a.nhead + head];
  const int ntile = (klen + BT - 1) / BT;

  // ---- sweep 1: L = log2 sum_j exp2(c s_j) per query (lane = query; this wave sees key blocks kb, kb + 2, ...) ----
  float m_run = -INFINITY, l_run = 0.f;
  {
    TileRegs rk = tile_fetch(a.k, a.ks, kbeg, klen, 0, hoff);
    tile_store(Ks[0], rk);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
      const int buf = it & 1;
      const bool more = it + 1 < ntile;
      if (more) rk = tile_fetch(a.k, a.ks, kbeg, klen, (it + 1) * BT, hoff);
      const f32x16 s = mm_rows(Ks[buf], 32 * kb, l31, h, qreg);       // S^T[key][query]
      float x[16], mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = it * BT + 32 * kb + crow(r, h);
        x[r] = key < klen ? s[r] * c : -INFINITY;
        mx = fmaxf(mx, x[r]);
      }
      if (mx > -INFINITY) {
        const float m_new = fmaxf(m_run, mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(x[r] - m_new);
        l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + sum;
        m_run = m_new;
      }
      if (more) tile_store(Ks[buf ^ 1], rk);
      __syncthreads();
    }
  }
  // merge the two half-waves (same query, different key rows), then the two key-block waves
  {
    const float m_o = __shfl_xor(m_run, 32, 64), l_o = __shfl_xor(l_run, 32, 64);
    const float m_n = fmaxf(m_run, m_o);
    float l_n = 0.f;
    if (m_run > -INFINITY) l_n += l_run * __builtin_amdgcn_exp2f(m_run - m_n);
    if (m_o > -INFINITY) l_n += l_o * __builtin_amdgcn_exp2f(m_o - m_n);
    m_run = m_n;
    l_run = l_n;
  }
  if (h == 0) {
    red_m[kb][32 * qb + l31] = m_run;
    red_l[kb][32 * qb + l31] = l_run;
  }
  __syncthreads();
  float lse;
  {
    const float m0 = red_m[0][32 * qb + l31], m1 = red_m[1][32 * qb + l31];
    const float l0 = red_l[0][32 * qb + l31], l1 = red_l[1][32 * qb + l31];
    const float m_n = fmaxf(m0, m1);
    float l_n = 0.f;
    if (m0 > -INFINITY) l_n += l0 * __builtin_amdgcn_exp2f(m0 - m_n);
    if (m1 > -INFINITY) l_n += l1 * __builtin_amdgcn_exp2f(m1 - m_n);
    lse = m_n + __builtin_amdgcn_logf(l_n);     // v_log_f32 = log2
  }
  if (kb == 0 && h == 0 && qvalid) a.lse[(size_t)(qbeg + qi) * a.nhead + head] = lse;
  __syncthreads();

  // ---- sweep 2: dQ^T[d][query] = sum_keys K^T dS^T ----
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    TileRegs rk = tile_fetch(a.k, a.ks, kbeg, klen, 0, hoff);
    TileRegs rv = tile_fetch(a.v, a.vs, kbeg, klen, 0, hoff);
    tile_store(Ks[0], rk);
    tile_store(Vs[0], rv);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
      const int buf = it & 1;
      const bool more = it + 1 < ntile;
      if (more) {
        rk = tile_fetch(a.k, a.ks, kbeg, klen, (it + 1) * BT, hoff);
        rv = tile_fetch(a.v, a.vs, kbeg, klen, (it + 1) * BT, hoff);
      }
      const f32x16 s = mm_rows(Ks[buf], 32 * kb, l31, h, qreg);       // S^T
      const f32x16 dp = mm_rows(Vs[buf], 32 * kb, l31, h, doreg);     // dP^T = V dO^T
      f32x16 ds;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = it * BT + 32 * kb + crow(r, h);
        const float p = key < klen ? __builtin_amdgcn_exp2f(s[r] * c - lse) : 0.f;
        ds[r] = p * (dp[r] - dq_row);
      }
      mm_cols_acc(acc, Ks[buf], 32 * kb, l31, h, ds);
      if (more) {
        tile_store(Ks[buf ^ 1], rk);
        tile_store(Vs[buf ^ 1], rv);
      }
      __syncthreads();
    }
  }
  // sum the two key-block waves in a fixed order, scale, store: lane (query l31, h) holds d = 8 a + 4 h + b
  if (kb == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red_acc[qb][r][lane] = acc[r];
  }
  __syncthreads();
  if (kb == 0 && qvalid) {
    float* dst = a.dq + (size_t)(qbeg + qi) * (a.nhead * BHD) + hoff + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 o;
      o.x = (acc[4 * g] + red_acc[qb][4 * g][lane]) * a.scale;
      o.y = (acc[4 * g + 1] + red_acc[qb][4 * g + 1][lane]) * a.scale;
      o.z = (acc[4 * g + 2] + red_acc[qb][4 * g + 2][lane]) * a.scale;
      o.w = (acc[4 * g + 3] + red_acc[qb][4 * g + 3][lane]) * a.scale;
      *reinterpret_cast<float4*>(dst + 8 * g) = o;
    }
  }
}

// grid (key tile, head, key segment)
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(const AttnBwdArgs a) {
  __shared__ __align__(16) float Qs[2][BT * LS], Os[2][BT * LS];
  __shared__ __align__(16) float Lt[2][BT], Dt[2][BT];
  __shared__ float red_acc[2][2][16][64];
  const int head = blockIdx.y, ksg = blockIdx.z;
  const int kbeg = a.cu[ksg], klen = a.cu[ksg + 1] - kbeg;
  const int k0 = blockIdx.x * BT;
  if (k0 >= klen) return;
  const int seg = a.q_seg[ksg];
  const int qbeg = a.cu[seg], qlen = a.cu[seg + 1] - qbeg;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int qb = wave & 1, kb = wave >> 1;
  const int hoff = head * BHD;
  const float c = a.scale * kLog2e;

  const int ki = k0 + 32 * kb + l31;
  const bool kvalid = ki < klen;
  const int kic = kvalid ? ki : klen - 1;
  float kreg[16], vreg[16];
  {
    const float4* pk = reinterpret_cast<const float4*>(a.k + (size_t)(kbeg + kic) * a.ks + hoff + 16 * h);
    const float4* pv = reinterpret_cast<const float4*>(a.v + (size_t)(kbeg + kic) * a.vs + hoff + 16 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 x = pk[j], y = pv[j];
      kreg[4 * j] = x.x; kreg[4 * j + 1] = x.y; kreg[4 * j + 2] = x.z; kreg[4 * j + 3] = x.w;
      vreg[4 * j] = y.x; vreg[4 * j + 1] = y.y; vreg[4 * j + 2] = y.z; vreg[4 * j + 3] = y.w;
    }
  }
  f32x16 acc_v, acc_k;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_v[r] = acc_k[r] = 0.f;
  const int ntile = (qlen + BT - 1) / BT;
  float rl = 0.f, rd = 0.f;
  auto fetch_stats = [&](int row0) {
    if (tid < BT) {
      const int r = row0 + tid;
      rl = r < qlen ? a.lse[(size_t)(qbeg + r) * a.nhead + head] : 0.f;
      rd = r < qlen ? a.dsum[(size_t)(qbeg + r) * a.nhead + head] : 0.f;
    }
  };
  auto store_stats = [&](int buf) {
    if (tid < BT) {
      Lt[buf][tid] = rl;
      Dt[buf][tid] = rd;
    }
  };
  TileRegs rq = tile_fetch(a.q, a.qs, qbeg, qlen, 0, hoff);
  TileRegs ro = tile_fetch(a.dout, a.dos, qbeg, qlen, 0, hoff);
  fetch_stats(0);
  tile_store(Qs[0], rq);
  tile_store(Os[0], ro);
  store_stats(0);
  __syncthreads();
  for (int it = 0; it < ntile; ++it) {
    const int buf = it & 1;
    const bool more = it + 1 < ntile;
    if (more) {
      rq = tile_fetch(a.q, a.qs, qbeg, qlen, (it + 1) * BT, hoff);
      ro = tile_fetch(a.dout, a.dos, qbeg, qlen, (it + 1) * BT, hoff);
      fetch_stats((it + 1) * BT);
    }
    const f32x16 s = mm_rows(Qs[buf], 32 * qb, l31, h, kreg);      // S[query][key]
    const f32x16 dp = mm_rows(Os[buf], 32 * qb, l31, h, vreg);     // dP = dO V^T
    f32x16 p, ds;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(&Lt[buf][32 * qb + 8 * g + 4 * h]);
      const float4 d4 = *reinterpret_cast<const float4*>(&Dt[buf][32 * qb + 8 * g + 4 * h]);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int r = 4 * g + b;
        const int qrow = it * BT + 32 * qb + 8 * g + 4 * h + b;
        const float pv = (qrow < qlen && kvalid) ? __builtin_amdgcn_exp2f(s[r] * c - lv[b]) : 0.f;
        p[r] = pv;
        ds[r] = pv * (dp[r] - dv[b]);
      }
    }
    mm_cols_acc(acc_v, Os[buf], 32 * qb, l31, h, p);      // dV^T[d][key] += dO^T P
    mm_cols_acc(acc_k, Qs[buf], 32 * qb, l31, h, ds);     // dK^T[d][key] += Q^T dS
    if (more) {
      tile_store(Qs[buf ^ 1], rq);
      tile_store(Os[buf ^ 1], ro);
      store_stats(buf ^ 1);
    }
    __syncthreads();
  }
  if (qb == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red_acc[0][kb][r][lane] = acc_v[r];
      red_acc[1][kb][r][lane] = acc_k[r];
    }
  }
  __syncthreads();
  if (qb == 0 && kvalid) {
    float* dv_dst = a.dv + (size_t)(kbeg + ki) * (a.nhead * BHD) + hoff + 4 * h;
    float* dk_dst = a.dk + (size_t)(kbeg + ki) * (a.nhead * BHD) + hoff + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 o, w;
      o.x = acc_v[4 * g] + red_acc[0][kb][4 * g][lane];
      o.y = acc_v[4 * g + 1] + red_acc[0][kb][4 * g + 1][lane];
      o.z = acc_v[4 * g + 2] + red_acc[0][kb][4 * g + 2][lane];
      o.w = acc_v[4 * g + 3] + red_acc[0][kb][4 * g + 3][lane];
      w.x = (acc_k[4 * g] + red_acc[1][kb][4 * g][lane]) * a.scale;
      w.y = (acc_k[4 * g + 1] + red_acc[1][kb][4 * g + 1][lane]) * a.scale;
      w.z = (acc_k[4 * g + 2] + red_acc[1][kb][4 * g + 2][lane]) * a.scale;
      w.w = (acc_k[4 * g + 3] + red_acc[1][kb][4 * g + 3][lane]) * a.scale;
      *reinterpret_cast<float4*>(dv_dst + 8 * g) = o;
      *reinterpret_cast<float4*>(dk_dst + 8 * g) = w;
    }
  }
}


// ======================================================================================================
// Split-fp16 form (attention mode 1, the default): the same two kernels with every product on
// v_mfma_f32_32x32x16_f16 over range-scaled hi/lo operand planes (3 matrix instructions per product and k-step, fp32
// accumulation -- the forward's arithmetic, spr_common.h split_pk_s), 5.3x less matrix time than the f32 form.
//   operand scales (powers of two, measured per call): sq, sk, sv, sdo bring max|q|, |k|, |v|, |dO| into
//   [2^14, 2^15).  S' = sq sk S and dP' = sv sdo dP stay in fp32.  The probabilities are split as P 2^14 (<= 2^14),
//   dS as P (dP' - D') 2^-21: |dP' - D'| <= 2 . 32 . 2^15 . 2^15 = 2^36 by Cauchy-Schwarz (D' is a convex
//   combination of the dP' of its row), so no operand can overflow fp16 whatever the data.
// The staged tiles live in LDS as fp16 planes in two forms: R[row][d] (A operand of the products that contract
// over d: 8 consecutive d per lane) and C[d][pos(row)] (A operand of the products that contract over the tile's
// rows), pos() = the order in which a 32x32 accumulator tile holds its rows, so that the accumulator registers
// of S / dS are the B operand of the next product as they are: k-step s of a 32-row block contracts the rows
// 16 s + 8 (j >> 2) + 4 h + (j & 3), j = 0..7, of lane half h = registers 8 s + j.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int bu32x2 __attribute__((ext_vector_type(2)));
constexpr int RS = 40;                    // R-form row stride in halves (80 B: 16-byte aligned, spreads the banks)
constexpr int CS = 72;                    // C-form row stride in halves (144 B)
constexpr int R_HALVES = BT * RS;         // one R plane of a 64-row tile
constexpr int C_HALVES = BHD * CS;        // one C plane
constexpr float P_MUL = 16384.f;          // 2^14
constexpr float DS_MUL = 4.76837158203125e-07f;   // 2^-21

__device__ __forceinline__ int cpos(int rho) {     // position of row rho (0..31) of a block in the C form
  return 16 * (rho >> 4) + 8 * ((rho >> 2) & 1) + 4 * ((rho >> 3) & 1) + (rho & 3);
}
// stores the fetched [64 x 32] tile into the R planes (and the C planes when ch != nullptr), scaled by s
__device__ __forceinline__ void tile_store_split(const TileRegs& t, float s, _Float16* rh, _Float16* rl, _Float16* ch,
                                                 _Float16* cl) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float4 v = e == 0 ? t.a : t.b;
    const int idx = threadIdx.x + 256 * e;
    const int row = idx >> 3, dc = 4 * (idx & 7);
    unsigned int h0, l0, h1, l1;
    split_pk_s(v.x, v.y, s, h0, l0);
    split_pk_s(v.z, v.w, s, h1, l1);
    *reinterpret_cast<bu32x2*>(rh + row * RS + dc) = (bu32x2){h0, h1};
    *reinterpret_cast<bu32x2*>(rl + row * RS + dc) = (bu32x2){l0, l1};
    if (ch != nullptr) {
      const int pos = 32 * (row >> 5) + cpos(row & 31);
      unsigned short* ph = reinterpret_cast<unsigned short*>(ch) + dc * CS + pos;
      unsigned short* pl = reinterpret_cast<unsigned short*>(cl) + dc * CS + pos;
      ph[0] = (unsigned short)(h0 & 0xffffu);
      ph[CS] = (unsigned short)(h0 >> 16);
      ph[2 * CS] = (unsigned short)(h1 & 0xffffu);
      ph[3 * CS] = (unsigned short)(h1 >> 16);
      pl[0] = (unsigned short)(l0 & 0xffffu);
      pl[CS] = (unsigned short)(l0 >> 16);
      pl[2 * CS] = (unsigned short)(l1 & 0xffffu);
      pl[3 * CS] = (unsigned short)(l1 >> 16);
    }
  }
}
// the lane's 16 values x[16 h' ..] of a row as B-operand planes: b?[s] = x[16 s + 8 h + j], j = 0..7
__device__ __forceinline__ void row_planes(const float* __restrict__ row, int h, float s, h16x8 (&bh)[2], h16x8 (&bl)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const float4 x = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * h);
    const float4 y = *reinterpret_cast<const float4*>(row + 16 * ks + 8 * h + 4);
    unsigned int hi[4], lo[4];
    split_pk_s(x.x, x.y, s, hi[0], lo[0]);
    split_pk_s(x.z, x.w, s, hi[1], lo[1]);
    split_pk_s(y.x, y.y, s, hi[2], lo[2]);
    split_pk_s(y.z, y.w, s, hi[3], lo[3]);
    bh[ks] = __builtin_bit_cast(h16x8, (bu32x4){hi[0], hi[1], hi[2], hi[3]});
    bl[ks] = __builtin_bit_cast(h16x8, (bu32x4){lo[0], lo[1], lo[2], lo[3]});
  }
}
__device__ __forceinline__ f32x16 mfma_h(const h16x8& a, const h16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// C = X[row0 + m][.] . B (contraction over d), X in R form
__device__ __forceinline__ f32x16 mm_rows_h(const _Float16* rh, const _Float16* rl, int row0, int l31, int h,
                                            const h16x8 (&bh)[2], const h16x8 (&bl)[2]) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const h16x8 ah = *reinterpret_cast<const h16x8*>(rh + (row0 + l31) * RS + 16 * ks + 8 * h);
    const h16x8 al = *reinterpret_cast<const h16x8*>(rl + (row0 + l31) * RS + 16 * ks + 8 * h);
    c = mfma_h(ah, bl[ks], c);
    c = mfma_h(al, bh[ks], c);
    c = mfma_h(ah, bh[ks], c);
  }
  return c;
}
// the accumulator tile b (scaled by s) as B-operand planes of a product that contracts over its rows
__device__ __forceinline__ void acc_planes(const f32x16& b, float s, h16x8 (&bh)[2], h16x8 (&bl)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    unsigned int hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split_pk_s(b[8 * ks + 2 * i], b[8 * ks + 2 * i + 1], s, hi[i], lo[i]);
    bh[ks] = __builtin_bit_cast(h16x8, (bu32x4){hi[0], hi[1], hi[2], hi[3]});
    bl[ks] = __builtin_bit_cast(h16x8, (bu32x4){lo[0], lo[1], lo[2], lo[3]});
  }
}
// acc += X^T . B, X in C form (32-row block blk), B = acc_planes of a C-layout tile
__device__ __forceinline__ void mm_cols_acc_h(f32x16& acc, const _Float16* ch, const _Float16* cl, int blk, int l31, int h,
                                              const h16x8 (&bh)[2], const h16x8 (&bl)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const h16x8 ah = *reinterpret_cast<const h16x8*>(ch + l31 * CS + 32 * blk + 16 * ks + 8 * h);
    const h16x8 al = *reinterpret_cast<const h16x8*>(cl + l31 * CS + 32 * blk + 16 * ks + 8 * h);
    acc = mfma_h(ah, bl[ks], acc);
    acc = mfma_h(al, bh[ks], acc);
    acc = mfma_h(ah, bh[ks], acc);
  }
}

// scales[0..3] = sq, sk, sv, sdo from the four max-|x| partial arrays
__global__ __launch_bounds__(256) void k_attn_bwd_scales(const float* __restrict__ parts, float* __restrict__ scales) {
  __shared__ float sh[17];
  for (int i = 0; i < 4; ++i) {
    const float m = block_absmax(parts + (size_t)i * kAmaxParts, sh, kAmaxParts);
    if (threadIdx.x == 0) scales[i] = pow2f(pow2_exp_for(m));
    __syncthreads();
  }
}

// ---- operand planes written once per call (k_attn_bwd_pack) -----------------------------------------------------
// Every workgroup of the two kernels stages 31 tiles of the OTHER side; converting them from fp32 on the way (the
// first form of this round) repeats the split and the transposing 2-byte LDS writes 31 times per tile.  With the
// planes in memory a staged tile is 16-byte copies: R form [head][token][32] hi / lo for q, k, v, dO; C form
// [head][d][column] hi / lo for q, k, dO with column = cst[segment] + 64 tile + pos(row) -- segments start at
// multiples of 64 columns (cst), rows past a segment's end are stored as zeros.
struct BwdPlanes {
  _Float16 *rq[2], *rk[2], *rv[2], *ro[2];
  _Float16 *cq[2], *ck[2], *co[2];
  int* cst;       // [nseg + 1]
  int t, tc;      // tokens, columns of a C plane row
};
struct Stage {
  TileRegs f;                  // fp32 route
  bu32x4 rh, rl, ch, cl;       // plane route
};
template <bool PL>
__device__ __forceinline__ void stage_fetch(Stage& st, const float* __restrict__ x, int stride, int hoff,
                                            _Float16* const (&rp)[2], _Float16* const (&cp)[2], bool want_c, int T,
                                            int tc, int head, int seg_beg, int len, int row0, int ccol0) {
  if constexpr (!PL) {
    st.f = tile_fetch(x, stride, seg_beg, len, row0, hoff);
  } else {
    const int row = row0 + (threadIdx.x >> 2), chunk = threadIdx.x & 3;
    st.rh = st.rl = (bu32x4){0u, 0u, 0u, 0u};
    if (row < len) {
      const size_t o = ((size_t)head * T + seg_beg + row) * BHD + 8 * chunk;
      st.rh = *reinterpret_cast<const bu32x4*>(rp[0] + o);
      st.rl = *reinterpret_cast<const bu32x4*>(rp[1] + o);
    }
    if (want_c) {
      const int d = threadIdx.x >> 3, c8 = threadIdx.x & 7;
      const size_t o = ((size_t)head * BHD + d) * tc + ccol0 + 8 * c8;
      st.ch = *reinterpret_cast<const bu32x4*>(cp[0] + o);
      st.cl = *reinterpret_cast<const bu32x4*>(cp[1] + o);
    }
  }
}
template <bool PL>
__device__ __forceinline__ void stage_store(const Stage& st, float s, _Float16* rh, _Float16* rl, _Float16* ch,
                                            _Float16* cl) {
  if constexpr (!PL) {
    tile_store_split(st.f, s, rh, rl, ch, cl);
  } else {
    const int row = threadIdx.x >> 2, chunk = threadIdx.x & 3;
    *reinterpret_cast<bu32x4*>(rh + row * RS + 8 * chunk) = st.rh;
    *reinterpret_cast<bu32x4*>(rl + row * RS + 8 * chunk) = st.rl;
    if (ch != nullptr) {
      const int d = threadIdx.x >> 3, c8 = threadIdx.x & 7;
      *reinterpret_cast<bu32x4*>(ch + d * CS + 8 * c8) = st.ch;
      *reinterpret_cast<bu32x4*>(cl + d * CS + 8 * c8) = st.cl;
    }
  }
}
// the lane's B-operand planes of one row, from the fp32 tensor (split here) or from the R planes
template <bool PL>
__device__ __forceinline__ void row_operand(const float* __restrict__ x, size_t xoff, float s, _Float16* const (&rp)[2],
                                            size_t prow, int h, h16x8 (&bh)[2], h16x8 (&bl)[2]) {
  if constexpr (!PL) {
    row_planes(x + xoff, h, s, bh, bl);
  } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bh[ks] = *reinterpret_cast<const h16x8*>(rp[0] + prow * BHD + 16 * ks + 8 * h);
      bl[ks] = *reinterpret_cast<const h16x8*>(rp[1] + prow * BHD + 16 * ks + 8 * h);
    }
  }
}

// first C column of every segment
__global__ void k_attn_bwd_cst(const int* __restrict__ cu, int nseg, int* __restrict__ cst) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int c = 0;
  for (int s = 0; s < nseg; ++s) {
    cst[s] = c;
    c += (cu[s + 1] - cu[s] + BT - 1) / BT * BT;
  }
  cst[nseg] = c;
}

// grid (tile, head, 4 segment + tensor): splits one [64 x 32] tile of q / k / v / dO into its planes
__global__ __launch_bounds__(256) void k_attn_bwd_pack(const AttnBwdArgs a, const BwdPlanes pl,
                                                       const float* __restrict__ scales) {
  __shared__ __align__(16) _Float16 th[BHD * CS], tl[BHD * CS];
  const int which = blockIdx.z & 3, seg = blockIdx.z >> 2, head = blockIdx.y;
  const int beg = a.cu[seg], len = a.cu[seg + 1] - beg;
  const int row0 = blockIdx.x * BT;
  if (row0 >= len) return;
  const float* x = which == 0 ? a.q : which == 1 ? a.k : which == 2 ? a.v : a.dout;
  const int stride = which == 0 ? a.qs : which == 1 ? a.ks : which == 2 ? a.vs : a.dos;
  _Float16* const* rp = which == 0 ? pl.rq : which == 1 ? pl.rk : which == 2 ? pl.rv : pl.ro;
  _Float16* const* cp = which == 0 ? pl.cq : which == 1 ? pl.ck : pl.co;
  const float s = scales[which];
  const int row = threadIdx.x >> 2, chunk = threadIdx.x & 3;
  const int r = row0 + row;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (r < len) {
    const float4* src = reinterpret_cast<const float4*>(x + (size_t)(beg + r) * stride + head * BHD + 8 * chunk);
    v0 = src[0];
    v1 = src[1];
  }
  unsigned int hi[4], lo[4];
  split_pk_s(v0.x, v0.y, s, hi[0], lo[0]);
  split_pk_s(v0.z, v0.w, s, hi[1], lo[1]);
  split_pk_s(v1.x, v1.y, s, hi[2], lo[2]);
  split_pk_s(v1.z, v1.w, s, hi[3], lo[3]);
  if (r < len) {
    const size_t o = ((size_t)head * pl.t + beg + r) * BHD + 8 * chunk;
    *reinterpret_cast<bu32x4*>(rp[0] + o) = (bu32x4){hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<bu32x4*>(rp[1] + o) = (bu32x4){lo[0], lo[1], lo[2], lo[3]};
  }
  if (which == 2) return;      // V is only ever contracted over d
  const int pos = 32 * (row >> 5) + cpos(row & 31);
  unsigned short* ph = reinterpret_cast<unsigned short*>(th) + (8 * chunk) * CS + pos;
  unsigned short* pw = reinterpret_cast<unsigned short*>(tl) + (8 * chunk) * CS + pos;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ph[(2 * i) * CS] = (unsigned short)(hi[i] & 0xffffu);
    ph[(2 * i + 1) * CS] = (unsigned short)(hi[i] >> 16);
    pw[(2 * i) * CS] = (unsigned short)(lo[i] & 0xffffu);
    pw[(2 * i + 1) * CS] = (unsigned short)(lo[i] >> 16);
  }
  __syncthreads();
  const int d = threadIdx.x >> 3, c8 = threadIdx.x & 7;
  const size_t o = ((size_t)head * BHD + d) * pl.tc + pl.cst[seg] + row0 + 8 * c8;
  *reinterpret_cast<bu32x4*>(cp[0] + o) = *reinterpret_cast<const bu32x4*>(th + d * CS + 8 * c8);
  *reinterpret_cast<bu32x4*>(cp[1] + o) = *reinterpret_cast<const bu32x4*>(tl + d * CS + 8 * c8);
}

#ifndef SPR_ATTN_BWD_NB
#define SPR_ATTN_BWD_NB 1
#endif
constexpr int NB = SPR_ATTN_BWD_NB;       // tile buffers.  1 (default): two barriers per iteration, half the LDS, THREE
                                          // workgroups per CU (measured 127.7 vs 131.0 ms per training step against 2)
constexpr int DQ_BUF_HALVES = 4 * R_HALVES + 2 * C_HALVES;     // K: R + C planes, V: R planes
constexpr int DKV_BUF_HALVES = 4 * R_HALVES + 4 * C_HALVES;    // Q and dO: R + C planes

// grid (query tile, head, query segment)
// HL: the per-query log-sum-exp comes from the forward (a.lse, spr_attn_varlen_fwd_lse): no sweep 1
template <bool PL, bool HL>
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void k_attn_bwd_dq_h(const AttnBwdArgs a, const float* __restrict__ scales,
                                                          const BwdPlanes pl) {
  __shared__ __align__(16) _Float16 tiles[NB * DQ_BUF_HALVES];
  __shared__ float red_m[2][BT], red_l[2][BT];
  const int head = blockIdx.y, seg = blockIdx.z;
  const int qbeg = a.cu[seg], qlen = a.cu[seg + 1] - qbeg;
  const int q0 = blockIdx.x * BT;
  if (q0 >= qlen) return;
  const int ksg = a.kv_seg[seg];
  const int kbeg = a.cu[ksg], klen = a.cu[ksg + 1] - kbeg;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int qb = wave & 1, kb = wave >> 1;
  const int hoff = head * BHD;
  const float sq = scales[0], sk = scales[1], sv = scales[2], sdo = scales[3];
  const float c = a.scale * kLog2e / (sq * sk);
  auto KRh = [&](int b) { return tiles + b * DQ_BUF_HALVES; };
  auto KRl = [&](int b) { return tiles + b * DQ_BUF_HALVES + R_HALVES; };
  auto VRh = [&](int b) { return tiles + b * DQ_BUF_HALVES + 2 * R_HALVES; };
  auto VRl = [&](int b) { return tiles + b * DQ_BUF_HALVES + 3 * R_HALVES; };
  auto KCh = [&](int b) { return tiles + b * DQ_BUF_HALVES + 4 * R_HALVES; };
  auto KCl = [&](int b) { return tiles + b * DQ_BUF_HALVES + 4 * R_HALVES + C_HALVES; };

  const int qi = q0 + 32 * qb + l31;
  const bool qvalid = qi < qlen;
  const int qic = qvalid ? qi : qlen - 1;
  h16x8 qh[2], ql[2], doh[2], dol[2];
  row_operand<PL>(a.q, (size_t)(qbeg + qic) * a.qs + hoff, sq, pl.rq, (size_t)head * pl.t + qbeg + qic, h, qh, ql);
  row_operand<PL>(a.dout, (size_t)(qbeg + qic) * a.dos + hoff, sdo, pl.ro, (size_t)head * pl.t + qbeg + qic, h, doh, dol);
  const int kc0 = PL ? pl.cst[ksg] : 0;
  auto fetchK = [&](Stage& st, int row0, bool want_c) __attribute__((always_inline)) {
    stage_fetch<PL>(st, a.k, a.ks, hoff, pl.rk, pl.ck, want_c, pl.t, pl.tc, head, kbeg, klen, row0, kc0 + row0);
  };
  auto fetchV = [&](Stage& st, int row0) __attribute__((always_inline)) {
    stage_fetch<PL>(st, a.v, a.vs, hoff, pl.rv, pl.rv, false, pl.t, pl.tc, head, kbeg, klen, row0, 0);
  };
  const float dq_row = a.dsum[(size_t)(qbeg + qic) * a.nhead + head] * (sv * sdo);
  const int ntile = (klen + BT - 1) / BT;

  float lse;
  if constexpr (HL) {
    lse = a.lse[(size_t)(qbeg + qic) * a.nhead + head];
  } else {
  // ---- sweep 1: L = log2 sum_j exp2(c s_j) per query ----
  float m_run = -INFINITY, l_run = 0.f;
  {
    Stage rk;
    fetchK(rk, 0, false);
    stage_store<PL>(rk, sk, KRh(0), KRl(0), nullptr, nullptr);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
      const int buf = it & (NB - 1), nbuf = (it + 1) & (NB - 1);
      const bool more = it + 1 < ntile;
      if (more) fetchK(rk, (it + 1) * BT, false);
      const f32x16 s = mm_rows_h(KRh(buf), KRl(buf), 32 * kb, l31, h, qh, ql);       // S'^T[key][query]
      float x[16], mx = -INFINITY;
      if (more) {                 // only the last tile of a segment can hold rows past its end
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          x[r] = s[r] * c;
          mx = fmaxf(mx, x[r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = it * BT + 32 * kb + crow(r, h);
          x[r] = key < klen ? s[r] * c : -INFINITY;
          mx = fmaxf(mx, x[r]);
        }
      }
      if (mx > -INFINITY) {
        const float m_new = fmaxf(m_run, mx);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(x[r] - m_new);
        l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + sum;
        m_run = m_new;
      }
      if (NB == 1) __syncthreads();
      if (more) stage_store<PL>(rk, sk, KRh(nbuf), KRl(nbuf), nullptr, nullptr);
      __syncthreads();
    }
  }
  {
    const float m_o = __shfl_xor(m_run, 32, 64), l_o = __shfl_xor(l_run, 32, 64);
    const float m_n = fmaxf(m_run, m_o);
    float l_n = 0.f;
    if (m_run > -INFINITY) l_n += l_run * __builtin_amdgcn_exp2f(m_run - m_n);
    if (m_o > -INFINITY) l_n += l_o * __builtin_amdgcn_exp2f(m_o - m_n);
    m_run = m_n;
    l_run = l_n;
  }
  if (h == 0) {
    red_m[kb][32 * qb + l31] = m_run;
    red_l[kb][32 * qb + l31] = l_run;
  }
  __syncthreads();
  {
    const float m0 = red_m[0][32 * qb + l31], m1 = red_m[1][32 * qb + l31];
    const float l0 = red_l[0][32 * qb + l31], l1 = red_l[1][32 * qb + l31];
    const float m_n = fmaxf(m0, m1);
    float l_n = 0.f;
    if (m0 > -INFINITY) l_n += l0 * __builtin_amdgcn_exp2f(m0 - m_n);
    if (m1 > -INFINITY) l_n += l1 * __builtin_amdgcn_exp2f(m1 - m_n);
    lse = m_n + __builtin_amdgcn_logf(l_n);     // v_log_f32 = log2
  }
  if (kb == 0 && h == 0 && qvalid) a.lse[(size_t)(qbeg + qi) * a.nhead + head] = lse;
  __syncthreads();

  }
  // ---- sweep 2: dQ^T[d][query] = sum_keys K^T dS^T ----
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    Stage rk, rv;
    fetchK(rk, 0, true);
    fetchV(rv, 0);
    stage_store<PL>(rk, sk, KRh(0), KRl(0), KCh(0), KCl(0));
    stage_store<PL>(rv, sv, VRh(0), VRl(0), nullptr, nullptr);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
      const int buf = it & (NB - 1), nbuf = (it + 1) & (NB - 1);
      const bool more = it + 1 < ntile;
      if (more) {
        fetchK(rk, (it + 1) * BT, true);
        fetchV(rv, (it + 1) * BT);
      }
      const f32x16 s = mm_rows_h(KRh(buf), KRl(buf), 32 * kb, l31, h, qh, ql);       // S'^T
      const f32x16 dp = mm_rows_h(VRh(buf), VRl(buf), 32 * kb, l31, h, doh, dol);    // dP'^T = V dO^T
      f32x16 ds;
      if (more) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(s[r] * c - lse) * (dp[r] - dq_row);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = it * BT + 32 * kb + crow(r, h);
          const float p = key < klen ? __builtin_amdgcn_exp2f(s[r] * c - lse) : 0.f;
          ds[r] = p * (dp[r] - dq_row);
        }
      }
      h16x8 bh[2], bl[2];
      acc_planes(ds, DS_MUL, bh, bl);
      mm_cols_acc_h(acc, KCh(buf), KCl(buf), kb, l31, h, bh, bl);
      if (NB == 1) __syncthreads();
      if (more) {
        stage_store<PL>(rk, sk, KRh(nbuf), KRl(nbuf), KCh(nbuf), KCl(nbuf));
        stage_store<PL>(rv, sv, VRh(nbuf), VRl(nbuf), nullptr, nullptr);
      }
      __syncthreads();
    }
  }
  // sum the two key-block waves in a fixed order, unscale, store: lane (query l31, h) holds d = 8 a + 4 h + b
  float (*red_acc)[16][64] = reinterpret_cast<float (*)[16][64]>(tiles);      // [2][16][64] floats, tiles are done
  if (kb == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red_acc[qb][r][lane] = acc[r];
  }
  __syncthreads();
  if (kb == 0 && qvalid) {
    const float u0 = 2097152.f / sk, u1 = a.scale / (sv * sdo);     // 1 / (sk . sv sdo 2^-21), in two exact steps
    float* dst = a.dq + (size_t)(qbeg + qi) * (a.nhead * BHD) + hoff + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 o;
      o.x = (acc[4 * g] + red_acc[qb][4 * g][lane]) * u0 * u1;
      o.y = (acc[4 * g + 1] + red_acc[qb][4 * g + 1][lane]) * u0 * u1;
      o.z = (acc[4 * g + 2] + red_acc[qb][4 * g + 2][lane]) * u0 * u1;
      o.w = (acc[4 * g + 3] + red_acc[qb][4 * g + 3][lane]) * u0 * u1;
      *reinterpret_cast<float4*>(dst + 8 * g) = o;
    }
  }
}

// grid (key tile, head, key segment); dynamic LDS: 2 x DKV_BUF_HALVES halves + 4 x 64 floats
template <bool PL>
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void k_attn_bwd_dkv_h(const AttnBwdArgs a, const float* __restrict__ scales,
                                                           const BwdPlanes pl) {
  extern __shared__ __align__(16) unsigned char dkv_smem[];
  _Float16* tiles = reinterpret_cast<_Float16*>(dkv_smem);
  float* Lt = reinterpret_cast<float*>(dkv_smem + (size_t)NB * DKV_BUF_HALVES * 2);     // [2][BT]
  float* Dt = Lt + 2 * BT;                                                               // [2][BT]
  const int head = blockIdx.y, ksg = blockIdx.z;
  const int kbeg = a.cu[ksg], klen = a.cu[ksg + 1] - kbeg;
  const int k0 = blockIdx.x * BT;
  if (k0 >= klen) return;
  const int seg = a.q_seg[ksg];
  const int qbeg = a.cu[seg], qlen = a.cu[seg + 1] - qbeg;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int qb = wave & 1, kb = wave >> 1;
  const int hoff = head * BHD;
  const float sq = scales[0], sk = scales[1], sv = scales[2], sdo = scales[3];
  const float c = a.scale * kLog2e / (sq * sk);
  const float dmul = sv * sdo;
  auto QRh = [&](int b) { return tiles + b * DKV_BUF_HALVES; };
  auto QRl = [&](int b) { return tiles + b * DKV_BUF_HALVES + R_HALVES; };
  auto ORh = [&](int b) { return tiles + b * DKV_BUF_HALVES + 2 * R_HALVES; };
  auto ORl = [&](int b) { return tiles + b * DKV_BUF_HALVES + 3 * R_HALVES; };
  auto QCh = [&](int b) { return tiles + b * DKV_BUF_HALVES + 4 * R_HALVES; };
  auto QCl = [&](int b) { return tiles + b * DKV_BUF_HALVES + 4 * R_HALVES + C_HALVES; };
  auto OCh = [&](int b) { return tiles + b * DKV_BUF_HALVES + 4 * R_HALVES + 2 * C_HALVES; };
  auto OCl = [&](int b) { return tiles + b * DKV_BUF_HALVES + 4 * R_HALVES + 3 * C_HALVES; };

  const int ki = k0 + 32 * kb + l31;
  const bool kvalid = ki < klen;
  const int kic = kvalid ? ki : klen - 1;
  h16x8 kh[2], kl[2], vh[2], vl[2];
  row_operand<PL>(a.k, (size_t)(kbeg + kic) * a.ks + hoff, sk, pl.rk, (size_t)head * pl.t + kbeg + kic, h, kh, kl);
  row_operand<PL>(a.v, (size_t)(kbeg + kic) * a.vs + hoff, sv, pl.rv, (size_t)head * pl.t + kbeg + kic, h, vh, vl);
  const int qc0 = PL ? pl.cst[seg] : 0;
  auto fetchQ = [&](Stage& st, int row0) __attribute__((always_inline)) {
    stage_fetch<PL>(st, a.q, a.qs, hoff, pl.rq, pl.cq, true, pl.t, pl.tc, head, qbeg, qlen, row0, qc0 + row0);
  };
  auto fetchO = [&](Stage& st, int row0) __attribute__((always_inline)) {
    stage_fetch<PL>(st, a.dout, a.dos, hoff, pl.ro, pl.co, true, pl.t, pl.tc, head, qbeg, qlen, row0, qc0 + row0);
  };
  f32x16 acc_v, acc_k;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_v[r] = acc_k[r] = 0.f;
  const int ntile = (qlen + BT - 1) / BT;
  float rl = 0.f, rd = 0.f;
  auto fetch_stats = [&](int row0) {
    if (tid < BT) {
      const int r = row0 + tid;
      rl = r < qlen ? a.lse[(size_t)(qbeg + r) * a.nhead + head] : 0.f;
      rd = r < qlen ? a.dsum[(size_t)(qbeg + r) * a.nhead + head] * dmul : 0.f;
    }
  };
  auto store_stats = [&](int buf) {
    if (tid < BT) {
      Lt[buf * BT + tid] = rl;
      Dt[buf * BT + tid] = rd;
    }
  };
  Stage rq, ro;
  fetchQ(rq, 0);
  fetchO(ro, 0);
  fetch_stats(0);
  stage_store<PL>(rq, sq, QRh(0), QRl(0), QCh(0), QCl(0));
  stage_store<PL>(ro, sdo, ORh(0), ORl(0), OCh(0), OCl(0));
  store_stats(0);
  __syncthreads();
  for (int it = 0; it < ntile; ++it) {
    const int buf = it & (NB - 1), nbuf = (it + 1) & (NB - 1);
    const bool more = it + 1 < ntile;
    if (more) {
      fetchQ(rq, (it + 1) * BT);
      fetchO(ro, (it + 1) * BT);
      fetch_stats((it + 1) * BT);
    }
    const f32x16 s = mm_rows_h(QRh(buf), QRl(buf), 32 * qb, l31, h, kh, kl);      // S'[query][key]
    const f32x16 dp = mm_rows_h(ORh(buf), ORl(buf), 32 * qb, l31, h, vh, vl);     // dP' = dO V^T
    f32x16 p, ds;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(&Lt[buf * BT + 32 * qb + 8 * g + 4 * h]);
      const float4 d4 = *reinterpret_cast<const float4*>(&Dt[buf * BT + 32 * qb + 8 * g + 4 * h]);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
      // (rows past the segment's end exist in the last tile only; an invalid key lane is never stored)
      if (more) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int r = 4 * g + b;
          const float pv = __builtin_amdgcn_exp2f(s[r] * c - lv[b]);
          p[r] = pv;
          ds[r] = pv * (dp[r] - dv[b]);
        }
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int r = 4 * g + b;
          const int qrow = it * BT + 32 * qb + 8 * g + 4 * h + b;
          const float pv = qrow < qlen ? __builtin_amdgcn_exp2f(s[r] * c - lv[b]) : 0.f;
          p[r] = pv;
          ds[r] = pv * (dp[r] - dv[b]);
        }
      }
    }
    {
      h16x8 bh[2], bl[2];
      acc_planes(p, P_MUL, bh, bl);
      mm_cols_acc_h(acc_v, OCh(buf), OCl(buf), qb, l31, h, bh, bl);      // dV^T[d][key] += dO^T P
      acc_planes(ds, DS_MUL, bh, bl);
      mm_cols_acc_h(acc_k, QCh(buf), QCl(buf), qb, l31, h, bh, bl);      // dK^T[d][key] += Q^T dS
    }
    if (NB == 1) __syncthreads();
    if (more) {
      stage_store<PL>(rq, sq, QRh(nbuf), QRl(nbuf), QCh(nbuf), QCl(nbuf));
      stage_store<PL>(ro, sdo, ORh(nbuf), ORl(nbuf), OCh(nbuf), OCl(nbuf));
      store_stats(nbuf);
    }
    __syncthreads();
  }
  float (*red_acc)[2][16][64] = reinterpret_cast<float (*)[2][16][64]>(dkv_smem);   // [2][2][16][64]: the tiles are done
  if (qb == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red_acc[0][kb][r][lane] = acc_v[r];
      red_acc[1][kb][r][lane] = acc_k[r];
    }
  }
  __syncthreads();
  if (qb == 0 && kvalid) {
    const float uv = 1.f / (sdo * P_MUL);
    const float u0 = 2097152.f / sq, u1 = a.scale / (sv * sdo);
    float* dv_dst = a.dv + (size_t)(kbeg + ki) * (a.nhead * BHD) + hoff + 4 * h;
    float* dk_dst = a.dk + (size_t)(kbeg + ki) * (a.nhead * BHD) + hoff + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 o, w;
      o.x = (acc_v[4 * g] + red_acc[0][kb][4 * g][lane]) * uv;
      o.y = (acc_v[4 * g + 1] + red_acc[0][kb][4 * g + 1][lane]) * uv;
      o.z = (acc_v[4 * g + 2] + red_acc[0][kb][4 * g + 2][lane]) * uv;
      o.w = (acc_v[4 * g + 3] + red_acc[0][kb][4 * g + 3][lane]) * uv;
      w.x = (acc_k[4 * g] + red_acc[1][kb][4 * g][lane]) * u0 * u1;
      w.y = (acc_k[4 * g + 1] + red_acc[1][kb][4 * g + 1][lane]) * u0 * u1;
      w.z = (acc_k[4 * g + 2] + red_acc[1][kb][4 * g + 2][lane]) * u0 * u1;
      w.w = (acc_k[4 * g + 3] + red_acc[1][kb][4 * g + 3][lane]) * u0 * u1;
      *reinterpret_cast<float4*>(dv_dst + 8 * g) = o;
      *reinterpret_cast<float4*>(dk_dst + 8 * g) = w;
    }
  }
}

}  // namespace
}  // namespace spr

using namespace spr;

static size_t attn_bwd_base_bytes(int t, int nhead) {
  // lse, dsum [t, nhead]; four max-|x| partial arrays and the four operand scales of the split-fp16 form
  return 2 * align_up((size_t)(t > 0 ? t : 1) * (size_t)(nhead > 0 ? nhead : 1) * sizeof(float), 256) +
         align_up((size_t)4 * kAmaxParts * sizeof(float), 256) + 256;
}
static int attn_bwd_tc(int t, int nseg) { return (t + BT - 1) / BT * BT + BT * (nseg > 0 ? nseg : 1); }
static size_t attn_bwd_plane_bytes(int t, int nseg, int nhead) {
  const size_t d = (size_t)(nhead > 0 ? nhead : 1) * BHD;
  return 8 * align_up((size_t)(t > 0 ? t : 1) * d * 2, 256) + 6 * align_up(d * (size_t)attn_bwd_tc(t, nseg) * 2, 256) +
         align_up((size_t)(nseg + 2) * sizeof(int), 256);
}
extern "C" size_t spr_attn_bwd_workspace_bytes(int t, int nhead) { return attn_bwd_base_bytes(t, nhead); }
// with room for the operand planes of the split-fp16 form (k_attn_bwd_pack); a workspace of only
// spr_attn_bwd_workspace_bytes still works -- the kernels then convert the fp32 tiles they stage themselves
extern "C" size_t spr_attn_bwd_workspace_bytes2(int t, int nseg, int nhead) {
  return attn_bwd_base_bytes(t, nhead) + attn_bwd_plane_bytes(t, nseg, nhead);
}

// q, k, v, out (the forward's output), dout: [t, nhead * 32] with unit inner stride and the given row strides;
// kv_seg: key segment of every query segment -- must be a permutation; q_seg: its inverse.
// dq, dk, dv: [t, nhead * 32] contiguous, fully written.
static int attn_varlen_bwd_impl(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                               int v_stride, const float* out, int o_stride, const float* dout, int do_stride,
                               const float* lse_in, const int* cu, const int* kv_seg, const int* q_seg, int t, int nseg,
                               int max_len_host, int nhead, int head_dim, float scale, float* dq, float* dk,
                               float* dv, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(head_dim == BHD, "attn_bwd: head_dim must be 32 (got %d)", head_dim);
  SPR_REQUIRE(q && k && v && out && dout && cu && kv_seg && q_seg && dq && dk && dv, "attn_bwd: null operand");
  SPR_REQUIRE(t > 0 && nseg >= 1 && nhead >= 1 && max_len_host >= 1, "attn_bwd: bad sizes");
  const int d = nhead * BHD;
  SPR_REQUIRE(q_stride >= d && k_stride >= d && v_stride >= d && o_stride >= d && do_stride >= d &&
                  q_stride % 4 == 0 && k_stride % 4 == 0 && v_stride % 4 == 0 && o_stride % 4 == 0 && do_stride % 4 == 0,
              "attn_bwd: row strides must be multiples of 4 floats and >= nhead * 32");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_attn_bwd_workspace_bytes(t, nhead), "attn_bwd: workspace too small");
  Workspace w(ws, ws_bytes);
  AttnBwdArgs a;
  a.lse = w.take<float>((size_t)t * nhead);
  a.dsum = w.take<float>((size_t)t * nhead);
  SPR_REQUIRE(a.dsum != nullptr, "attn_bwd: workspace carve failed");
  // the forward's log-sum-exp (split-fp16 forms only: the exact-f32 kernels keep their own sweep)
  const bool have_lse = lse_in != nullptr && attn_mode() != 0;
  if (have_lse) a.lse = const_cast<float*>(lse_in);
  a.q = q; a.k = k; a.v = v; a.out = out; a.dout = dout;
  a.qs = q_stride; a.ks = k_stride; a.vs = v_stride; a.os = o_stride; a.dos = do_stride;
  a.cu = cu; a.kv_seg = kv_seg; a.q_seg = q_seg; a.nseg = nseg; a.nhead = nhead; a.scale = scale;
  a.dq = dq; a.dk = dk; a.dv = dv;
  hipLaunchKernelGGL(k_attn_bwd_rowdot, dim3(cdiv((long)t * nhead, 256)), dim3(256), 0, stream, out, o_stride, dout,
                     do_stride, t, nhead, a.dsum);
  const dim3 grid(cdiv(max_len_host, BT), nhead, nseg);
  if (attn_mode() != 0) {
    // split-fp16 form: operand scales from the measured maxima of q, k, v, dO
    float* parts = w.take<float>((size_t)4 * kAmaxParts);
    float* scales = w.take<float>(4);
    SPR_REQUIRE(scales != nullptr, "attn_bwd: workspace carve failed");
    if (int rc = launch_absmax2(q, t, d, q_stride, parts, k, t, d, k_stride, parts + kAmaxParts, stream)) return rc;
    if (int rc = launch_absmax2(v, t, d, v_stride, parts + 2 * kAmaxParts, dout, t, d, do_stride, parts + 3 * kAmaxParts,
                                stream))
      return rc;
    hipLaunchKernelGGL(k_attn_bwd_scales, dim3(1), dim3(256), 0, stream, parts, scales);
    constexpr size_t dkv_lds = (size_t)NB * DKV_BUF_HALVES * 2 + 4 * BT * sizeof(float);
    static_assert(dkv_lds >= sizeof(float) * 2 * 2 * 16 * 64, "the final reduction reuses the tile buffers");
    static const bool no_planes = getenv("SPR_ATTN_BWD_PLANES") != nullptr && getenv("SPR_ATTN_BWD_PLANES")[0] == '0';
    BwdPlanes pl{};
    if (!no_planes && ws_bytes >= spr_attn_bwd_workspace_bytes2(t, nseg, nhead)) {
      // operand planes written once (k_attn_bwd_pack), staged by 16-byte copies
      const int tc = attn_bwd_tc(t, nseg);
      _Float16** r[4] = {pl.rq, pl.rk, pl.rv, pl.ro};
      for (auto& rp : r)
        for (int i = 0; i < 2; ++i) rp[i] = w.take<_Float16>((size_t)t * d);
      _Float16** cpl[3] = {pl.cq, pl.ck, pl.co};
      for (auto& cp : cpl)
        for (int i = 0; i < 2; ++i) cp[i] = w.take<_Float16>((size_t)d * tc);
      pl.cst = w.take<int>(nseg + 1);
      pl.t = t;
      pl.tc = tc;
      SPR_REQUIRE(pl.cst != nullptr, "attn_bwd: workspace carve failed");
      hipLaunchKernelGGL(k_attn_bwd_cst, dim3(1), dim3(64), 0, stream, cu, nseg, pl.cst);
      hipLaunchKernelGGL(k_attn_bwd_pack, dim3(cdiv(max_len_host, BT), nhead, 4 * nseg), dim3(256), 0, stream, a, pl, scales);
      if (have_lse)
        hipLaunchKernelGGL((k_attn_bwd_dq_h<true, true>), grid, dim3(256), 0, stream, a, scales, pl);
      else
        hipLaunchKernelGGL((k_attn_bwd_dq_h<true, false>), grid, dim3(256), 0, stream, a, scales, pl);
      static const int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_dkv_h<true>),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)dkv_lds);
      SPR_REQUIRE(attr_rc == 0, "attn_bwd: cannot reserve %zu bytes of LDS", dkv_lds);
      hipLaunchKernelGGL(k_attn_bwd_dkv_h<true>, grid, dim3(256), dkv_lds, stream, a, scales, pl);
      SPR_LAUNCH_CHECK();
      return 0;
    }
    if (have_lse)
      hipLaunchKernelGGL((k_attn_bwd_dq_h<false, true>), grid, dim3(256), 0, stream, a, scales, pl);
    else
      hipLaunchKernelGGL((k_attn_bwd_dq_h<false, false>), grid, dim3(256), 0, stream, a, scales, pl);
    static const int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_dkv_h<false>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dkv_lds);
    SPR_REQUIRE(attr_rc == 0, "attn_bwd: cannot reserve %zu bytes of LDS", dkv_lds);
    hipLaunchKernelGGL(k_attn_bwd_dkv_h<false>, grid, dim3(256), dkv_lds, stream, a, scales, pl);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_attn_bwd_dq, grid, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_attn_bwd_dkv, grid, dim3(256), 0, stream, a);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_attn_varlen_bwd(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                                   int v_stride, const float* out, int o_stride, const float* dout, int do_stride,
                                   const int* cu, const int* kv_seg, const int* q_seg, int t, int nseg,
                                   int max_len_host, int nhead, int head_dim, float scale, float* dq, float* dk,
                                   float* dv, void* ws, size_t ws_bytes, void* stream_) {
  return attn_varlen_bwd_impl(q, q_stride, k, k_stride, v, v_stride, out, o_stride, dout, do_stride, nullptr, cu, kv_seg,
                              q_seg, t, nseg, max_len_host, nhead, head_dim, scale, dq, dk, dv, ws, ws_bytes, stream_);
}

// lse [t, nhead]: what spr_attn_varlen_fwd_lse handed out for the same q, k (NULL: computed here)
extern "C" int spr_attn_varlen_bwd_lse(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                                       int v_stride, const float* out, int o_stride, const float* dout, int do_stride,
                                       const float* lse, const int* cu, const int* kv_seg, const int* q_seg, int t,
                                       int nseg, int max_len_host, int nhead, int head_dim, float scale, float* dq,
                                       float* dk, float* dv, void* ws, size_t ws_bytes, void* stream_) {
  return attn_varlen_bwd_impl(q, q_stride, k, k_stride, v, v_stride, out, o_stride, dout, do_stride, lse, cu, kv_seg,
                              q_seg, t, nseg, max_len_host, nhead, head_dim, scale, dq, dk, dv, ws, ws_bytes, stream_);
}
