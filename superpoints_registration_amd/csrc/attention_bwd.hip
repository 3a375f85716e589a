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
// Arithmetic: exact f32 MFMA (v_mfma_f32_32x32x2_f32) like spr_bgemm, fp32 softmax with v_exp_f32;
// fixed summation order -> bitwise reproducible gradients.
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

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_attn_bwd_workspace_bytes(int t, int nhead) {
  return 2 * align_up((size_t)(t > 0 ? t : 1) * (size_t)(nhead > 0 ? nhead : 1) * sizeof(float), 256);
}

// q, k, v, out (the forward's output), dout: [t, nhead * 32] with unit inner stride and the given row strides;
// kv_seg: key segment of every query segment -- must be a permutation; q_seg: its inverse.
// dq, dk, dv: [t, nhead * 32] contiguous, fully written.
extern "C" int spr_attn_varlen_bwd(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                                   int v_stride, const float* out, int o_stride, const float* dout, int do_stride,
                                   const int* cu, const int* kv_seg, const int* q_seg, int t, int nseg,
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
  a.q = q; a.k = k; a.v = v; a.out = out; a.dout = dout;
  a.qs = q_stride; a.ks = k_stride; a.vs = v_stride; a.os = o_stride; a.dos = do_stride;
  a.cu = cu; a.kv_seg = kv_seg; a.q_seg = q_seg; a.nseg = nseg; a.nhead = nhead; a.scale = scale;
  a.dq = dq; a.dk = dk; a.dv = dv;
  hipLaunchKernelGGL(k_attn_bwd_rowdot, dim3(cdiv((long)t * nhead, 256)), dim3(256), 0, stream, out, o_stride, dout,
                     do_stride, t, nhead, a.dsum);
  const dim3 grid(cdiv(max_len_host, BT), nhead, nseg);
  hipLaunchKernelGGL(k_attn_bwd_dq, grid, dim3(256), 0, stream, a);
  hipLaunchKernelGGL(k_attn_bwd_dkv, grid, dim3(256), 0, stream, a);
  SPR_LAUNCH_CHECK();
  return 0;
}
