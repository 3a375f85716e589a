// Dense projections, LayerNorm and the sine positional embedding on gfx950.
//
// Behaviour contract:
//   nn.Linear call sites: UnaryBlock.mlp (kpconv_blocks.py:549,:557),
//     feat_proj / overlap_predictor (qk_regtr_full.py:47,:85,:176,:248-249),
//     MultiheadAttention in_proj / out_proj and linear1 / linear2
//     (transformer/transformers.py:96-104, :198-238)
//   nn.LayerNorm + with_pos_embed  transformers.py:121, :196-197, :212-214
//   PositionEmbeddingCoordsSine.forward  transformer/position_embedding.py:29-50
//
// GEMM: out[M,N] = act(X[M,K] @ W[N,K]^T + bias + residual), both operands
// K-contiguous ("NT").  Two kernels, chosen by spr_set_gemm_mode:
//   k_gemm_nt_h3 (default)  split-fp16 operands (fp32-level accuracy), three
//     v_mfma_f32_32x32x16_f16 per k-step; 256x256 / 128x64 / 128x32 tiles; an
//     optional epilogue writes the attention operand planes (attn_planes.h)
//     instead of a row-major fp32 matrix;
//   k_gemm_nt               exact-f32 matrix cores (v_mfma_f32_32x32x2_f32: a
//     k-ordered fmaf chain), one 32x32 accumulator per wave, LDS row stride of
//     33 words (conflict-free fragment reads).
// In both the next K-slab is prefetched into registers (inline-asm loads +
// explicit s_waitcnt) while the current one feeds the MFMAs.
#include <atomic>

#include "attn_planes.h"
#include "spr_common.h"

namespace spr {
namespace {

constexpr int BK = 32;
constexpr int LDSK = BK + 1;

// Epilogue of one 32x32 accumulator tile (C/D layout, dtype independent:
// col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).  ACT / RES are
// compile-time so the 16 residual loads are issued back to back (one exposed
// round trip per tile, not one per element) and no per-element branch remains.
template <int ACT, bool RES>
__device__ __forceinline__ float store_tile(const f32x16& acc, int row0, int col, int M, int N,
                                            float bv, const float* __restrict__ residual,
                                            float* __restrict__ out, int lh, float unscale = 1.0f,
                                            const float* __restrict__ epi = nullptr) {
  if (col >= N) return 0.f;
  float vmax = 0.f;
  float res[16];
  if (RES) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(row0 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1);
      res[r] = residual[(size_t)row * N + col];
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    float v = acc[r] * unscale + bv;   // unscale: exact power of two (1 in the exact-f32 kernel)
    if (RES) v += res[r];
    if (ACT == SPR_ACT_RELU) v = fmaxf(v, 0.f);
    if (ACT == SPR_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    // matching-head epilogues (match_pose.hip): same float operations, in the same order, as
    // the separate k_scale / k_affinity passes over the score matrix they replace
    if (ACT == kEpiScale) v = v * epi[0];
    if (ACT == kEpiAffinity) {
      const float sc = fmaxf(v * epi[0], 0.f);
      v = -(sc - epi[1]) * epi[2];
    }
    if (row < M) {
      out[(size_t)row * N + col] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
  }
  return vmax;
}

// WM x WN waves, each a 32x32 tile.  256 threads when WM*WN == 4.
template <int WM, int WN, int ACT, bool RES>
__global__ __launch_bounds__(WM* WN * 64) void k_gemm_nt(
    const float* __restrict__ X, int M, int K, const float* __restrict__ Wt, int N,
    const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ out) {
  constexpr int BM = WM * 32, BN = WN * 32, NT = WM * WN * 64;
  constexpr int A_F4 = BM * BK / 4, B_F4 = BN * BK / 4;  // float4 per slab
  constexpr int A_PT = (A_F4 + NT - 1) / NT, B_PT = (B_F4 + NT - 1) / NT;
  __shared__ float As[BM * LDSK];
  __shared__ float Bs[BN * LDSK];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  // 1-D grid, XCD-swizzled: the column tiles of one row panel share an L2
  const int gx = (N + BN - 1) / BN;
  const int lid = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (lid / gx) * BM, n0 = (lid % gx) * BN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  // Next K-slab prefetch: hipcc sinks plain loads to their first use (after the
  // MFMA block), which exposes a full memory round trip per slab, so the slab is
  // fetched with inline-asm loads and retired with an explicit s_waitcnt; no
  // other vector-memory operation is in flight inside the K loop.
  f32x4 ra[A_PT], rb[B_PT];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int f = tid + i * NT;
      if (f < A_F4) {   // compile-time true except for partial last pass
        const int r = f / (BK / 4), c4 = f % (BK / 4);
        const float* p = X + (size_t)min(m0 + r, M - 1) * K + k0 + c4 * 4;   // clamped row
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[i]) : "v"(p));
      }
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int f = tid + i * NT;
      if (f < B_F4) {
        const int r = f / (BK / 4), c4 = f % (BK / 4);
        const float* p = Wt + (size_t)min(n0 + r, N - 1) * K + k0 + c4 * 4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[i]) : "v"(p));
      }
    }
  };
  auto store_slab = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int f = tid + i * NT;
      if (f < A_F4) {
        const int r = f / (BK / 4), c4 = f % (BK / 4);
        const bool in = m0 + r < M;                     // rows past M contribute zeros
        float* d = As + r * LDSK + c4 * 4;
        d[0] = in ? ra[i][0] : 0.f;
        d[1] = in ? ra[i][1] : 0.f;
        d[2] = in ? ra[i][2] : 0.f;
        d[3] = in ? ra[i][3] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int f = tid + i * NT;
      if (f < B_F4) {
        const int r = f / (BK / 4), c4 = f % (BK / 4);
        const bool in = n0 + r < N;
        float* d = Bs + r * LDSK + c4 * 4;
        d[0] = in ? rb[i][0] : 0.f;
        d[1] = in ? rb[i][1] : 0.f;
        d[2] = in ? rb[i][2] : 0.f;
        d[3] = in ? rb[i][3] : 0.f;
      }
    }
  };

  load_slab(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    store_slab();
    __syncthreads();
    if (k0 + BK < K) load_slab(k0 + BK);
    __builtin_amdgcn_sched_barrier(0);
    const float* ap = As + (wm * 32 + l31) * LDSK + lh;
    const float* bp = Bs + (wn * 32 + l31) * LDSK + lh;
    // all fragments of the slab first (one exposed LDS latency), then 16 MFMAs
    float fa[BK / 2], fb[BK / 2];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      fa[s] = ap[2 * s];   // A[i = l&31][k = 2s + (l>>5)]
      fb[s] = bp[2 * s];   // B[k][j = l&31]
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < BK / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int col = n0 + wn * 32 + l31;
  const float bv = (bias && col < N) ? bias[col] : 0.f;
  store_tile<ACT, RES>(acc, m0 + wm * 32, col, M, N, bv, residual, out, lh);
}

// ---------------------------------------------------------------------------
// Split-fp16 GEMM ("h3"): fp32-level accuracy at ~5x the exact-f32 MFMA rate.
//   x' = x s_x (s_x = per-tensor power of two, max |x'| in [2^14, 2^15))
//   x' = hi + lo,  hi = fp16(x'),  lo = fp16(x' - hi)
//   a.b ~= (ah.bh + ah.bl + al.bh) / (s_a s_b)        (al.bl ~ 2^-22 |a.b| dropped)
// The matrix cores honour fp16 subnormals on both operands (measured with
// scripts/abl/denorm.hip), so all three products share ONE fp32 accumulator.
// Range safety (spr_common.h, split_pk_s): max |x| of each operand comes from a
// pre-pass (launch_absmax -> 512 partials, reduced in this kernel's prologue), so
// no operand can overflow fp16 and every element within 2^-18 of its tensor's
// maximum keeps 22 significand bits; smaller ones keep an absolute error of
// 2^-39 max|x|.  Products are exact in the MFMA and accumulate in fp32.  Three
// v_mfma_f32_32x32x16_f16 per 16-deep k-step (32 cycles each) replace eight
// v_mfma_f32_32x32x2_f32 (64 cycles each).  Operands are scaled + split on the
// fly while a K-slab is staged into LDS (fp16 rows of 32+8 halves: 80-byte stride
// -> conflict-free ds_read_b128 fragment reads).  spr_set_gemm_mode(0) selects the
// exact-f32 kernel.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int HS = 40;                 // LDS row stride in halves (BK = 32 + 8 pad)

template <int BM, int BN, int WM, int WN, int ACT, bool RES, bool PLANES = false>
__global__ __launch_bounds__(WM* WN * 64) void k_gemm_nt_h3(
    const float* __restrict__ Xp, int Mp, int K, const float* __restrict__ Wp, int Np,
    const float* __restrict__ bias, const float* __restrict__ residual,
    float* __restrict__ outp, AttnPlanes pl, int f0, const float* __restrict__ a_parts, int n_aparts,
    const float* __restrict__ w_parts, float* __restrict__ out_parts,
    const GemmGroup* __restrict__ groups, int n_wparts) {
  constexpr int NT = WM * WN * 64;      // threads (4 or 8 waves)
  // grouped launch (groups != nullptr): the grid is the concatenation of the tile grids of
  // independent problems X_g [m_g, K] x W_g [n_g, K]^T that share K and the operand scales
  // (the per-pair correlation GEMMs of the matching head: one launch instead of one per pair)
  const float* __restrict__ X = Xp;
  const float* __restrict__ Wt = Wp;
  float* __restrict__ out = outp;
  int M = Mp, N = Np, bid = blockIdx.x, nblk = gridDim.x;
  if (groups != nullptr) {
    int g = 0, beg = 0;
    while (bid >= groups[g].tile_end) beg = groups[g++].tile_end;
    X += groups[g].a_off;
    Wt += groups[g].b_off;
    out += groups[g].c_off;
    M = groups[g].m;
    N = groups[g].n;
    nblk = groups[g].tile_end - beg;
    bid -= beg;
  }
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);   // 32x32 sub-tiles per wave
  constexpr int A_F4 = BM * BK / 4, B_F4 = BN * BK / 4;
  constexpr int A_PT = A_F4 / NT, B_PT = B_F4 / NT;
  static_assert(A_F4 % NT == 0 && B_F4 % NT == 0, "slab must divide over the threads");
  extern __shared__ __align__(16) unsigned char gemm_smem[];
  _Float16* Ah = reinterpret_cast<_Float16*>(gemm_smem);
  _Float16* Al = Ah + BM * HS;
  _Float16* Bh = Al + BM * HS;
  _Float16* Bl = Bh + BN * HS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  // 1-D grid, XCD-swizzled: the column tiles of one row panel share an L2
  const int gx = (N + BN - 1) / BN;
  const int lid = groups != nullptr ? bid : xcd_swizzle(bid, nblk);
  const int m0 = (lid / gx) * BM, n0 = (lid % gx) * BN;
  const int l31 = lane & 31, lh = lane >> 5;

  // per-tensor power-of-two operand scales from the absmax partials
  float sa, sb, unscale;
  {
    float* shf = reinterpret_cast<float*>(gemm_smem);
    const int ka = pow2_exp_for(block_absmax(a_parts, shf, n_aparts));
    const int kb = pow2_exp_for(block_absmax(w_parts, shf, n_wparts));
    __syncthreads();   // shf aliases the operand tiles
    sa = pow2f(ka);
    sb = pow2f(kb);
    unscale = pow2f(-ka - kb);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[A_PT], rb[B_PT];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      const float* p = X + (size_t)min(m0 + r, M - 1) * K + k0 + c4 * 4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[i]) : "v"(p));
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      const float* p = Wt + (size_t)min(n0 + r, N - 1) * K + k0 + c4 * 4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[i]) : "v"(p));
    }
  };
  // rows past M / N are clamped on the load side: they produce finite garbage
  // in accumulator rows / columns that store_tile never writes
  auto split_store = [&](const f32x4& v, float sc, _Float16* hi, _Float16* lo) {
    unsigned int h0, h1, l0, l1;
    split_pk_s(v[0], v[1], sc, h0, l0);
    split_pk_s(v[2], v[3], sc, h1, l1);
    *reinterpret_cast<u32x2*>(hi) = (u32x2){h0, h1};
    *reinterpret_cast<u32x2*>(lo) = (u32x2){l0, l1};
  };
  auto store_slab = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      split_store(ra[i], sa, Ah + r * HS + c4 * 4, Al + r * HS + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      split_store(rb[i], sb, Bh + r * HS + c4 * 4, Bl + r * HS + c4 * 4);
    }
  };

  load_slab(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    store_slab();
    __syncthreads();
    if (k0 + BK < K) load_slab(k0 + BK);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      // lane: row (l&31) of its sub-tile, k = 16 s + 8 (l>>5) + 0..7
      const int ko = 16 * s + 8 * lh;
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + l31;
        ah[i] = *reinterpret_cast<const f16x8*>(Ah + row * HS + ko);
        al[i] = *reinterpret_cast<const f16x8*>(Al + row * HS + ko);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + l31;
        bh[j] = *reinterpret_cast<const f16x8*>(Bh + row * HS + ko);
        bl[j] = *reinterpret_cast<const f16x8*>(Bl + row * HS + ko);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // small terms first, then the dominant one
          if (PLANES) {
            // transposed product: lane = token (X row), registers = features (W rows)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[j], ah[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], al[i], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[j], ah[i], acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (PLANES) {
    // ---- attention operand planes (attn_planes.h) straight from the accumulators ----
    // accumulator (i, j): lane column = token m0 + (wm TM + i) 32 + l31, register r = feature
    // n0 + (wn TN + j) 32 + (r & 3) + 8 (r >> 2) + 4 lh of this launch (f0 + ... of the projection)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int tok = m0 + (wm * TM + i) * 32 + l31;
      if (tok >= M) continue;
      const int which = (f0 + n0) >> 8;                         // 0 Q, 1 K, 2 V (block uniform)
      // plane multiplier (power of two; Q additionally log2(e)/sqrt(d)) -- attention.hip,
      // k_plane_scales: keeps every plane inside fp16's range whatever the magnitudes
      const float pmul = pl.scales[which];
      int vcol = 0;
      if (which == 2) {
        const int sg = find_segment(pl.cu, pl.nseg, tok);
        vcol = attn_vperm(attn_vstart_of(pl.cu[sg], sg) + tok - pl.cu[sg]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int fl = n0 + (wn * TN + j) * 32;                 // first feature of the sub-tile (launch local)
        const int fg = (f0 + fl) & 255;                         // ... inside its Q / K / V block
        if (which == 2) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            // registers r, r+1 = adjacent features (two plane rows), same token column
            const int d = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float a = acc[i][j][r] * unscale + bias[fl + d];
            const float b = acc[i][j][r + 1] * unscale + bias[fl + d + 1];
            unsigned int hu, lu;
            split_pk_s(a, b, pmul, hu, lu);
            const f16x2 hh = __builtin_bit_cast(f16x2, hu), ll = __builtin_bit_cast(f16x2, lu);
            const size_t o = attn_v_off(fg + d, (size_t)vcol, 256);     // blocked planes: the next feature is 16 halves on
            pl.vth[o] = hh[0];
            pl.vth[o + 16] = hh[1];
            pl.vtl[o] = ll[0];
            pl.vtl[o + 16] = ll[1];
          }
        } else {
          _Float16* ph = which == 0 ? pl.qh : pl.kh;
          _Float16* pw = which == 0 ? pl.ql : pl.kl;
          const size_t row = ((size_t)(fg >> 5) * pl.t_total + tok) * 32;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d0 = 8 * g + 4 * lh;                      // registers 4g..4g+3 = features d0..d0+3
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * unscale + bias[fl + d0 + e];
            unsigned int h0, h1, l0, l1;
            split_pk_s(v[0], v[1], pmul, h0, l0);
            split_pk_s(v[2], v[3], pmul, h1, l1);
            *reinterpret_cast<u32x2*>(ph + row + d0) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2*>(pw + row + d0) = (u32x2){l0, l1};
          }
        }
      }
    }
    return;
  }
  float omax = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + l31;
    const float bv = (ACT < kEpiScale && bias && col < N) ? bias[col] : 0.f;   // epilogue modes: `bias` = parameters
#pragma unroll
    for (int i = 0; i < TM; ++i)
      omax = fmaxf(omax, store_tile<ACT, RES>(acc[i][j], m0 + (wm * TM + i) * 32, col, M, N, bv, residual, out, lh,
                                              unscale, ACT >= kEpiScale ? bias : nullptr));
  }
  if (out_parts != nullptr) {
    // publish max |out| of this tile: the consumer GEMM then needs no pass over `out` to scale it
    float* shf = reinterpret_cast<float*>(gemm_smem);
    omax = wave_max(omax);
    __syncthreads();
    if (lane == 0) shf[wave] = omax;
    __syncthreads();
    if (tid == 0) {
      float t = shf[0];
      for (int w_ = 1; w_ < WM * WN; ++w_) t = fmaxf(t, shf[w_]);
      out_parts[blockIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------------------
// Persistent form of k_gemm_nt_h3 for K = 256 whose output stores overlap the next tile's main loop.
// A 256 x 256 tile of k_gemm_nt_h3 ends with 256 KB of stores that drain at the CU's share of the
// HBM write rate (~11 us) with nothing else to do, and the next tile's operand loads queue behind
// them in the CU's one vector-memory pipeline: main loop and epilogue ADD (70 + 54 us for the
// [61 745 x 256] x [768 x 256] in-projection).  Here one workgroup per CU walks 128 x 256 tiles;
// a finished tile stays in registers (64 accumulators per thread) while the next tile's K loop
// runs into a second set, and its stores leave in sixteen "quarters" (4 accumulator registers of
// one 32 x 32 block = 4 consecutive rows of the lane's column), two per K slab.  The counted
// s_waitcnt of the slab prefetch leaves exactly those youngest stores outstanding (vector-memory
// operations retire in issue order), so a wave never waits for its own stores; the first slab of
// the NEXT tile is prefetched in the last iteration of the current one for the same reason.
// K slabs are unrolled (8), which makes every quarter's registers static.
template <int ACT>
__global__ __launch_bounds__(512) void k_gemm_nt_h3p(
    const float* __restrict__ X, int M, const float* __restrict__ Wt, int N,
    const float* __restrict__ bias, float* __restrict__ out,
    const float* __restrict__ a_parts, int n_aparts, const float* __restrict__ w_parts, int n_wparts,
    float* __restrict__ out_parts, int ntiles) {
  constexpr int K = 256, NSLAB = K / BK;
  constexpr int BM = 128, BN = 256, WN = 4, NT = 512, TM = 2, TN = 2;
  constexpr int A_PT = BM * BK / 4 / NT, B_PT = BN * BK / 4 / NT;   // 2, 4 float4 per thread and slab
  constexpr int QPS = 16 / NSLAB;                                   // quarters leaving per slab: 2
  extern __shared__ __align__(16) unsigned char gemm_smem[];
  _Float16* Ah = reinterpret_cast<_Float16*>(gemm_smem);
  _Float16* Al = Ah + BM * HS;
  _Float16* Bh = Al + BM * HS;
  _Float16* Bl = Bh + BN * HS;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int gx = (N + BN - 1) / BN;

  float sa, sb, unscale;
  {
    float* shf = reinterpret_cast<float*>(gemm_smem);
    const int ka = pow2_exp_for(block_absmax(a_parts, shf, n_aparts));
    const int kb = pow2_exp_for(block_absmax(w_parts, shf, n_wparts));
    __syncthreads();
    sa = pow2f(ka);
    sb = pow2f(kb);
    unscale = pow2f(-ka - kb);
  }

  f32x16 acc[TM][TN], prev[TM][TN];
  f32x4 ra[A_PT], rb[B_PT];
  auto load_slab = [&](int m0, int n0, int k0) {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      const float* p = X + (size_t)min(m0 + r, M - 1) * K + k0 + c4 * 4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[i]) : "v"(p));
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      const float* p = Wt + (size_t)min(n0 + r, N - 1) * K + k0 + c4 * 4;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[i]) : "v"(p));
    }
  };
  auto split_store = [&](const f32x4& v, float sc, _Float16* hi, _Float16* lo) {
    unsigned int h0, h1, l0, l1;
    split_pk_s(v[0], v[1], sc, h0, l0);
    split_pk_s(v[2], v[3], sc, h1, l1);
    *reinterpret_cast<u32x2*>(hi) = (u32x2){h0, h1};
    *reinterpret_cast<u32x2*>(lo) = (u32x2){l0, l1};
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      split_store(ra[i], sa, Ah + r * HS + c4 * 4, Al + r * HS + c4 * 4);
    }
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
      const int f = tid + i * NT;
      const int r = f / (BK / 4), c4 = f % (BK / 4);
      split_store(rb[i], sb, Bh + r * HS + c4 * 4, Bl + r * HS + c4 * 4);
    }
  };

  // deferred tile
  bool have_prev = false;
  int pm0 = 0, pn0 = 0, ptile = 0;
  float pbias[TN] = {0.f, 0.f};
  float pmax = 0.f;
  // quarter Q of prev: block Q >> 2 = (row block Q >> 3, column block (Q >> 2) & 1), registers 4 (Q & 3) .. +3
  auto emit = [&](auto qtag) {
    constexpr int Q = decltype(qtag)::value;
    constexpr int b = Q >> 2, g = Q & 3, bi = b >> 1, bj = b & 1;
    const int row0 = pm0 + (wm * TM + bi) * 32 + 8 * g + 4 * lh;
    const int col = pn0 + (wn * TN + bj) * 32 + l31;
    if (col < N) {
      float* p = out + (size_t)row0 * N + col;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = prev[bi][bj][4 * g + e] * unscale + pbias[bj];
        if (ACT == SPR_ACT_RELU) y = fmaxf(y, 0.f);
        if (ACT == SPR_ACT_SIGMOID) y = 1.f / (1.f + expf(-y));
        if (row0 + e < M) {
          asm volatile("global_store_dword %0, %1, off" ::"v"(p + (size_t)e * N), "v"(y) : "memory");
          pmax = fmaxf(pmax, fabsf(y));
        }
      }
    }
  };
  auto publish_prev_max = [&]() {   // all 512 threads, LDS not in use at the call sites
    if (out_parts == nullptr) return;
    float* shf = reinterpret_cast<float*>(gemm_smem);
    const float t_ = wave_max(pmax);
    __syncthreads();
    if (lane == 0) shf[wave] = t_;
    __syncthreads();
    if (tid == 0) {
      float m_ = shf[0];
      for (int w_ = 1; w_ < 8; ++w_) m_ = fmaxf(m_, shf[w_]);
      out_parts[ptile] = m_;
    }
    __syncthreads();
    pmax = 0.f;
  };

  int t = blockIdx.x;
  if (t < ntiles) {
    const int lid = xcd_swizzle(t, ntiles);
    load_slab((lid / gx) * BM, (lid % gx) * BN, 0);
  }
  bool young = false;   // the previous iteration issued its two quarters (8 stores) after the operand prefetch
  for (; t < ntiles; t += gridDim.x) {
    const int lid = xcd_swizzle(t, ntiles);
    const int m0 = (lid / gx) * BM, n0 = (lid % gx) * BN;
    const int tn = t + gridDim.x;
    int nm0 = 0, nn0 = 0;
    if (tn < ntiles) {
      const int nl = xcd_swizzle(tn, ntiles);
      nm0 = (nl / gx) * BM;
      nn0 = (nl % gx) * BN;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto slab = [&](auto stag) {
      constexpr int S = decltype(stag)::value;
      // operand slab S was prefetched one iteration ago; the 8 stores issued after it may stay in flight
      if (young) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      store_slab();
      __syncthreads();
      if (S + 1 < NSLAB) load_slab(m0, n0, (S + 1) * BK);
      else if (tn < ntiles) load_slab(nm0, nn0, 0);      // first slab of the next tile
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        const int ko = 16 * ks + 8 * lh;
        f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = (wm * TM + i) * 32 + l31;
          ah[i] = *reinterpret_cast<const f16x8*>(Ah + row * HS + ko);
          al[i] = *reinterpret_cast<const f16x8*>(Al + row * HS + ko);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = (wn * TN + j) * 32 + l31;
          bh[j] = *reinterpret_cast<const f16x8*>(Bh + row * HS + ko);
          bl[j] = *reinterpret_cast<const f16x8*>(Bl + row * HS + ko);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      young = false;
      if (have_prev) {
        emit(std::integral_constant<int, QPS * S>{});
        emit(std::integral_constant<int, QPS * S + 1>{});
        young = true;
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    };
    slab(std::integral_constant<int, 0>{});
    slab(std::integral_constant<int, 1>{});
    slab(std::integral_constant<int, 2>{});
    slab(std::integral_constant<int, 3>{});
    slab(std::integral_constant<int, 4>{});
    slab(std::integral_constant<int, 5>{});
    slab(std::integral_constant<int, 6>{});
    slab(std::integral_constant<int, 7>{});
    if (have_prev) publish_prev_max();
    // this tile becomes the deferred one
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) prev[i][j] = acc[i][j];
    pm0 = m0;
    pn0 = n0;
    ptile = t;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + l31;
      pbias[j] = (bias && col < N) ? bias[col] : 0.f;
    }
    have_prev = true;
  }
  if (have_prev) {   // the last tile of this workgroup: nothing left to hide its stores behind
    emit(std::integral_constant<int, 0>{});  emit(std::integral_constant<int, 1>{});
    emit(std::integral_constant<int, 2>{});  emit(std::integral_constant<int, 3>{});
    emit(std::integral_constant<int, 4>{});  emit(std::integral_constant<int, 5>{});
    emit(std::integral_constant<int, 6>{});  emit(std::integral_constant<int, 7>{});
    emit(std::integral_constant<int, 8>{});  emit(std::integral_constant<int, 9>{});
    emit(std::integral_constant<int, 10>{}); emit(std::integral_constant<int, 11>{});
    emit(std::integral_constant<int, 12>{}); emit(std::integral_constant<int, 13>{});
    emit(std::integral_constant<int, 14>{}); emit(std::integral_constant<int, 15>{});
    publish_prev_max();
  }
}

static std::atomic<int> g_gemm_mode{1};   // 1 = split-fp16 (default), 0 = exact f32 MFMA

// N small (overlap_predictor, N = 1): one wave per (row, n).
__global__ void k_gemv_rows(const float* __restrict__ X, int M, int K, const float* __restrict__ Wt,
                            int N, const float* __restrict__ bias,
                            const float* __restrict__ residual, int act, float* __restrict__ out) {
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wid >= (long)M * N) return;
  const int row = (int)(wid / N), n = (int)(wid % N);
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += X[(size_t)row * K + k] * Wt[(size_t)n * K + k];
  s = wave_sum(s);
  if (lane == 0) {
    float v = s + (bias ? bias[n] : 0.f);
    if (residual) v += residual[(size_t)row * N + n];
    if (act == SPR_ACT_RELU) v = fmaxf(v, 0.f);
    if (act == SPR_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    out[(size_t)row * N + n] = v;
  }
}

constexpr int kLnRangeSlots = kRangeSlots;
// One wave per row; c % 64 == 0, c <= 1024.
template <int MAXV>
__global__ void k_layernorm(const float* __restrict__ x, int m, int c,
                            const float* __restrict__ gamma, const float* __restrict__ beta,
                            float eps, const float* __restrict__ pos, float* __restrict__ out_norm,
                            float* __restrict__ out_pos, float* __restrict__ range_norm,
                            float* __restrict__ range_pos) {
  __shared__ float shr[8];
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  float mxn = 0.f, mxp = 0.f;
  if (row < m) {
  const int nv = c >> 6;
  float v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    v[i] = (i < nv) ? x[(size_t)row * c + i * 64 + lane] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)c;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const float d = (i < nv) ? v[i] - mean : 0.f;
    ss += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)c + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < nv) {
      const int ch = i * 64 + lane;
      const float y = (v[i] - mean) * rstd * gamma[ch] + beta[ch];
      mxn = fmaxf(mxn, fabsf(y));
      if (out_norm) out_norm[(size_t)row * c + ch] = y;
      if (out_pos) {
        const float yp = y + pos[(size_t)row * c + ch];
        mxp = fmaxf(mxp, fabsf(yp));
        out_pos[(size_t)row * c + ch] = yp;
      }
    }
  }
  }
  if (range_norm || range_pos) {
    // the consumer GEMM's operand range: kLnRangeSlots partial maxima, combined with integer atomic
    // max on the bit pattern (non-negative floats order like unsigned ints; order independent,
    // hence deterministic).  The caller zero-initialises the slots.
    mxn = wave_max(mxn);
    mxp = wave_max(mxp);
    if (lane == 0) {
      shr[threadIdx.x >> 6] = mxn;
      shr[4 + (threadIdx.x >> 6)] = mxp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int slot = blockIdx.x & (kLnRangeSlots - 1);
      if (range_norm)
        atomicMax(reinterpret_cast<unsigned int*>(range_norm) + slot,
                  __float_as_uint(fmaxf(fmaxf(shr[0], shr[1]), fmaxf(shr[2], shr[3]))));
      if (range_pos)
        atomicMax(reinterpret_cast<unsigned int*>(range_pos) + slot,
                  __float_as_uint(fmaxf(fmaxf(shr[4], shr[5]), fmaxf(shr[6], shr[7]))));
    }
  }
}

// c == 256 (d_model of every shipped config): one float4 per lane and tensor (1 KiB per wave
// load / store), kLnRows consecutive rows per wave with all their loads issued up front.
constexpr int kLnRows = 4;
__global__ __launch_bounds__(256) void k_layernorm256(const float* __restrict__ x, int m,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      const float* __restrict__ pos, float* __restrict__ out_norm,
                                                      float* __restrict__ out_pos, float* __restrict__ range_norm,
                                                      float* __restrict__ range_pos) {
  __shared__ float shr[8];
  const int row0 = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * kLnRows;
  const int lane = threadIdx.x & 63;
  float mxn = 0.f, mxp = 0.f;
  float4 v[kLnRows], pv[kLnRows];
#pragma unroll
  for (int r = 0; r < kLnRows; ++r) {
    if (row0 + r < m) {
      const size_t o = (size_t)(row0 + r) * 256 + 4 * lane;
      v[r] = *reinterpret_cast<const float4*>(x + o);
      if (out_pos) pv[r] = *reinterpret_cast<const float4*>(pos + o);
    }
  }
  const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  const float4 b = *reinterpret_cast<const float4*>(beta + 4 * lane);
#pragma unroll
  for (int r = 0; r < kLnRows; ++r) {
    if (row0 + r >= m) continue;
    const size_t o = (size_t)(row0 + r) * 256 + 4 * lane;
    const float mean = wave_sum((v[r].x + v[r].y) + (v[r].z + v[r].w)) / 256.0f;
    const float dx = v[r].x - mean, dy = v[r].y - mean, dz = v[r].z - mean, dw = v[r].w - mean;
    const float rstd = 1.0f / sqrtf(wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) / 256.0f + eps);
    float4 y;
    y.x = dx * rstd * g.x + b.x;
    y.y = dy * rstd * g.y + b.y;
    y.z = dz * rstd * g.z + b.z;
    y.w = dw * rstd * g.w + b.w;
    mxn = fmaxf(mxn, fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w))));
    if (out_norm) *reinterpret_cast<float4*>(out_norm + o) = y;
    if (out_pos) {
      float4 yp;
      yp.x = y.x + pv[r].x;
      yp.y = y.y + pv[r].y;
      yp.z = y.z + pv[r].z;
      yp.w = y.w + pv[r].w;
      mxp = fmaxf(mxp, fmaxf(fmaxf(fabsf(yp.x), fabsf(yp.y)), fmaxf(fabsf(yp.z), fabsf(yp.w))));
      *reinterpret_cast<float4*>(out_pos + o) = yp;
    }
  }
  if (range_norm || range_pos) {   // same hand-over as k_layernorm
    mxn = wave_max(mxn);
    mxp = wave_max(mxp);
    if (lane == 0) {
      shr[threadIdx.x >> 6] = mxn;
      shr[4 + (threadIdx.x >> 6)] = mxp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int slot = blockIdx.x & (kLnRangeSlots - 1);
      if (range_norm)
        atomicMax(reinterpret_cast<unsigned int*>(range_norm) + slot,
                  __float_as_uint(fmaxf(fmaxf(shr[0], shr[1]), fmaxf(shr[2], shr[3]))));
      if (range_pos)
        atomicMax(reinterpret_cast<unsigned int*>(range_pos) + slot,
                  __float_as_uint(fmaxf(fmaxf(shr[4], shr[5]), fmaxf(shr[6], shr[7]))));
    }
  }
}

// position_embedding.py:29-50
__global__ void k_posemb(const float* __restrict__ xyz, int n, int d_model, int npf, float scale,
                         float temperature, float* __restrict__ out) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * d_model) return;
  const int row = (int)(gid / d_model), ch = (int)(gid % d_model);
  float v = 0.f;
  if (ch < 3 * npf) {
    const int axis = ch / npf, i = ch % npf;
    // dim_t[i] = temperature ** (2 * (i // 2) / npf)
    const float e = (float)(2 * (i / 2)) / (float)npf;
    const float dim_t = powf(temperature, e);
    const float a = (xyz[3 * (size_t)row + axis] * scale) / dim_t;
    v = (i & 1) ? cosf(a) : sinf(a);
  }
  out[gid] = v;
}

}  // namespace
}  // namespace spr

using namespace spr;

namespace {
template <int ACT, bool RES>
int launch_gemm(const float* x, int m, int k, const float* w, int n, const float* bias,
                const float* residual, float* out, const float* a_parts, int n_aparts, const float* w_parts,
                int n_wparts, float* out_parts, int out_cap, int* out_n, hipStream_t stream) {
  if (out_n) *out_n = 0;
  if (spr::g_gemm_mode.load(std::memory_order_relaxed) == 1) {
    SPR_REQUIRE(a_parts != nullptr && w_parts != nullptr, "linear: split-fp16 mode needs the absmax partials");
    // LDS bytes of a BM x BN tile's slab (hi + lo planes of both operands)
    auto lds = [](int bm, int bn) { return (size_t)(bm + bn) * spr::HS * 2 * sizeof(_Float16); };
    static const bool use_p = [] { const char* e = getenv("SPR_GEMM_PERSIST"); return e == nullptr || e[0] != '0'; }();
    if (use_p && !RES && ACT <= SPR_ACT_SIGMOID && n >= 256 && k == 256 && m >= 1024 && m < 32768) {
      // persistent 128x256 tiles (k_gemm_nt_h3p): for a few thousand rows the finer tiles fill the
      // chip better (27 vs 43 us at 5 000 x 256 x 300); at 60 k rows both forms sit on the CU's
      // vector-memory throughput and the 256x256 tiles move fewer operand bytes
      auto kern = spr::k_gemm_nt_h3p<ACT>;
      if (int rc = ensure_dyn_lds((const void*)kern, (int)lds(128, 256))) return rc;
      const int ntiles = cdiv(n, 256) * cdiv(m, 128);
      int grid = device_cu_count();
      grid = grid < ntiles ? grid : ntiles;
      if (grid >= 8) grid &= ~7;                       // whole XCD rounds: t and t + grid share an L2
      float* op = (out_parts && ntiles <= out_cap) ? out_parts : nullptr;
      if (op && out_n) *out_n = ntiles;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds(128, 256), stream, x, m, w, n, bias, out,
                         a_parts, n_aparts, w_parts, n_wparts, op, ntiles);
    } else if (n >= 256 && m >= 256 && cdiv(n, 256) * cdiv(m, 256) >= 128) {
      // (fewer than 128 such tiles -- a training batch of 15 k tokens and N = 256 gives 61 -- leave most of the
      // chip idle: the 128x64 tiles below then win although they move more operand bytes)
      // 256x256 tiles, 8 waves of 64x128: ~64 flop per operand byte fetched from
      // L2 (the 128-wide tiles below need 2-4x the L2 traffic and are bound by it)
      auto kern = spr::k_gemm_nt_h3<256, 256, 4, 2, ACT, RES>;
      if (int rc = ensure_dyn_lds((const void*)kern, (int)lds(256, 256))) return rc;
      const int grid = cdiv(n, 256) * cdiv(m, 256);
      float* op = (out_parts && grid <= out_cap) ? out_parts : nullptr;   // range published only if it fits
      if (op && out_n) *out_n = grid;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds(256, 256), stream, x, m, k,
                         w, n, bias, residual, out, spr::AttnPlanes(), 0, a_parts, n_aparts, w_parts, op, (const spr::GemmGroup*)nullptr, n_wparts);
    } else if (n > 32) {
      const int grid = cdiv(n, 64) * cdiv(m, 128);
      float* op = (out_parts && grid <= out_cap) ? out_parts : nullptr;   // range published only if it fits
      if (op && out_n) *out_n = grid;
      hipLaunchKernelGGL((spr::k_gemm_nt_h3<128, 64, 4, 1, ACT, RES>), dim3(grid),
                         dim3(256), lds(128, 64), stream, x, m, k, w, n, bias, residual, out, spr::AttnPlanes(), 0,
                         a_parts, n_aparts, w_parts, op, (const spr::GemmGroup*)nullptr, n_wparts);
    } else {
      hipLaunchKernelGGL((spr::k_gemm_nt_h3<128, 32, 4, 1, ACT, RES>), dim3(cdiv(n, 32) * cdiv(m, 128)),
                         dim3(256), lds(128, 32), stream, x, m, k, w, n, bias, residual, out, spr::AttnPlanes(), 0,
                         a_parts, n_aparts, w_parts, (float*)nullptr, (const spr::GemmGroup*)nullptr, n_wparts);
    }
  } else if (n % 64 == 0) {
    hipLaunchKernelGGL((spr::k_gemm_nt<2, 2, ACT, RES>), dim3((n / 64) * cdiv(m, 64)), dim3(256), 0, stream,
                       x, m, k, w, n, bias, residual, out);
  } else {
    hipLaunchKernelGGL((spr::k_gemm_nt<4, 1, ACT, RES>), dim3(cdiv(n, 32) * cdiv(m, 128)), dim3(256), 0,
                       stream, x, m, k, w, n, bias, residual, out);
  }
  SPR_LAUNCH_CHECK();
  return 0;
}
}  // namespace

int spr::gemm_mode() { return spr::g_gemm_mode.load(std::memory_order_relaxed); }

int spr::launch_inproj_planes(const float* x, int m, int k, const float* w, int n, const float* bias, int f0,
                              const AttnPlanes& planes, const float* a_parts, int n_aparts, const float* w_parts,
                              hipStream_t stream) {
  SPR_REQUIRE(n % 256 == 0 && f0 % 256 == 0 && k % BK == 0 && m >= 1 && bias != nullptr,
              "in-projection planes: bad shape (m=%d k=%d n=%d)", m, k, n);
  SPR_REQUIRE(a_parts && w_parts && planes.scales, "in-projection planes: missing scale inputs");
  auto kern = spr::k_gemm_nt_h3<256, 256, 4, 2, SPR_ACT_NONE, false, true>;
  const size_t lds = (size_t)(256 + 256) * spr::HS * 2 * sizeof(_Float16);
  if (int rc = ensure_dyn_lds((const void*)kern, (int)lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(cdiv(n, 256) * cdiv(m, 256)), dim3(512), lds, stream, x, m, k, w, n, bias,
                     (const float*)nullptr, (float*)nullptr, planes, f0, a_parts, n_aparts, w_parts, (float*)nullptr,
                     (const spr::GemmGroup*)nullptr, kAmaxParts);
  SPR_LAUNCH_CHECK();
  return 0;
}

// GEMM with the operand ranges already measured (a_parts / w_parts: kAmaxParts partial
// maxima each, see launch_absmax); used by the matching head and the mode-0 fallbacks
// (which pass nullptr).
int spr::launch_linear_ranged(const float* x, int m, int k, const float* w, int n, const float* bias,
                              float* out, const float* a_parts, const float* w_parts, hipStream_t stream) {
  return launch_gemm<SPR_ACT_NONE, false>(x, m, k, w, n, bias, nullptr, out, a_parts, kAmaxParts, w_parts, kAmaxParts,
                                          nullptr, 0, nullptr, stream);
}

// Grouped NT GEMM, split-fp16 mode only: problem g is A_g = a + groups[g].a_off [m_g, k],
// B_g = b + groups[g].b_off [n_g, k], C_g = c + groups[g].c_off [m_g, n_g] (row-major, ld = n_g);
// groups[g].tile_end = running total of cdiv(m_g, bm) * cdiv(n_g, bn) with (bm, bn) =
// gemm_group_tile(max n_g).  One launch for all of them.
void spr::gemm_group_tile(int max_n, int* bm, int* bn) {
  if (max_n >= 256) { *bm = 256; *bn = 256; }
  else if (max_n > 32) { *bm = 128; *bn = 64; }
  else { *bm = 128; *bn = 32; }
}

namespace {
template <int ACT>
int launch_grouped_act(const float* a, int k, const float* b, float* c, const spr::GemmGroup* groups_dev,
                       int total_tiles, int max_n, const float* a_parts, const float* w_parts, const float* epi,
                       hipStream_t stream) {
  auto lds = [](int bm, int bn) { return (size_t)(bm + bn) * spr::HS * 2 * sizeof(_Float16); };
  int bm, bn;
  spr::gemm_group_tile(max_n, &bm, &bn);
  if (bn == 256) {
    auto kern = spr::k_gemm_nt_h3<256, 256, 4, 2, ACT, false>;
    if (int rc = ensure_dyn_lds((const void*)kern, (int)lds(256, 256))) return rc;
    hipLaunchKernelGGL(kern, dim3(total_tiles), dim3(512), lds(256, 256), stream, a, 0, k, b, 0, epi,
                       (const float*)nullptr, c, spr::AttnPlanes(), 0, a_parts, kAmaxParts, w_parts,
                       (float*)nullptr, groups_dev, kAmaxParts);
  } else if (bn == 64) {
    hipLaunchKernelGGL((spr::k_gemm_nt_h3<128, 64, 4, 1, ACT, false>), dim3(total_tiles), dim3(256),
                       lds(128, 64), stream, a, 0, k, b, 0, epi, (const float*)nullptr, c, spr::AttnPlanes(), 0,
                       a_parts, kAmaxParts, w_parts, (float*)nullptr, groups_dev, kAmaxParts);
  } else {
    hipLaunchKernelGGL((spr::k_gemm_nt_h3<128, 32, 4, 1, ACT, false>), dim3(total_tiles), dim3(256),
                       lds(128, 32), stream, a, 0, k, b, 0, epi, (const float*)nullptr, c, spr::AttnPlanes(), 0,
                       a_parts, kAmaxParts, w_parts, (float*)nullptr, groups_dev, kAmaxParts);
  }
  SPR_LAUNCH_CHECK();
  return 0;
}
}  // namespace

// epi_mode: 0 none, kEpiScale (out = v * epi[0]), kEpiAffinity (out = -(max(v epi[0], 0) - epi[1]) epi[2]);
// epi = device parameters of the epilogue.
int spr::launch_gemm_grouped(const float* a, int k, const float* b, float* c, const GemmGroup* groups_dev,
                             int total_tiles, int max_n, const float* a_parts, const float* w_parts,
                             int epi_mode, const float* epi, hipStream_t stream) {
  SPR_REQUIRE(spr::gemm_mode() == 1 && a_parts && w_parts && groups_dev && total_tiles >= 1 && k % BK == 0,
              "grouped gemm: bad arguments");
  SPR_REQUIRE(epi_mode == 0 || epi != nullptr, "grouped gemm: epilogue parameters missing");
  switch (epi_mode) {
    case 0: return launch_grouped_act<SPR_ACT_NONE>(a, k, b, c, groups_dev, total_tiles, max_n, a_parts, w_parts,
                                                    nullptr, stream);
    case kEpiScale: return launch_grouped_act<kEpiScale>(a, k, b, c, groups_dev, total_tiles, max_n, a_parts,
                                                         w_parts, epi, stream);
    case kEpiAffinity: return launch_grouped_act<kEpiAffinity>(a, k, b, c, groups_dev, total_tiles, max_n, a_parts,
                                                               w_parts, epi, stream);
  }
  SPR_REQUIRE(false, "grouped gemm: unknown epilogue %d", epi_mode);
  return 1;
}

extern "C" size_t spr_linear_workspace_bytes(void) { return 2 * align_up(kAmaxParts * sizeof(float), 256); }

extern "C" int spr_linear_r(const float* x, int m, int k, const float* w, int n, const float* bias,
                            const float* residual, int act, float* out, const float* x_range, int x_range_n,
                            const float* w_range, int w_range_n, float* out_range, int out_range_cap,
                            int* out_range_n_host, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (out_range_n_host) *out_range_n_host = 0;
  SPR_REQUIRE(m > 0 && k > 0 && n > 0, "linear: bad sizes m=%d k=%d n=%d", m, k, n);
  SPR_REQUIRE(act >= 0 && act <= 2, "linear: unknown activation %d", act);
  SPR_REQUIRE(x_range == nullptr || x_range_n >= 1, "linear: x_range needs a count");
  SPR_REQUIRE(w_range == nullptr || w_range_n >= 1, "linear: w_range needs a count");
  if (n < 16 || k % BK != 0) {
    SPR_REQUIRE(n <= 64 || k % BK == 0, "linear: k must be a multiple of %d for n > 64 (k=%d n=%d)", BK, k, n);
    const long waves = (long)m * n;
    hipLaunchKernelGGL(k_gemv_rows, dim3(cdiv(waves * 64, 256)), dim3(256), 0, stream, x, m, k, w, n,
                       bias, residual, act, out);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  const float *a_parts = nullptr, *w_parts = nullptr;
  int n_ap = kAmaxParts, n_wp = kAmaxParts;
  if (spr::gemm_mode() == 1) {
    SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_linear_workspace_bytes(),
                "linear: workspace too small (%zu bytes given, spr_linear_workspace_bytes() needed)", ws_bytes);
    Workspace wk(ws, ws_bytes);
    float* ap = wk.take<float>(kAmaxParts);
    float* wp = wk.take<float>(kAmaxParts);
    // ranges handed in (x: published by its producer; w: measured once per weight version by
    // the caller, spr_absmax) are not measured again -- with both there is no pre-pass at all
    a_parts = x_range != nullptr ? x_range : ap;
    w_parts = w_range != nullptr ? w_range : wp;
    if (x_range != nullptr) n_ap = x_range_n;
    if (w_range != nullptr) n_wp = w_range_n;
    if (x_range == nullptr && w_range == nullptr) {
      if (int rc = launch_absmax2(x, m, k, k, ap, w, n, k, k, wp, stream)) return rc;
    } else if (x_range == nullptr) {
      if (int rc = launch_absmax(x, m, k, k, ap, stream)) return rc;
    } else if (w_range == nullptr) {
      if (int rc = launch_absmax(w, n, k, k, wp, stream)) return rc;
    }
  }
  const bool res = residual != nullptr;
#define SPR_LG(A, R) launch_gemm<A, R>(x, m, k, w, n, bias, residual, out, a_parts, n_ap, w_parts, n_wp, out_range, \
                                       out_range_cap, out_range_n_host, stream)
  switch (act * 2 + (res ? 1 : 0)) {
    case 0: return SPR_LG(SPR_ACT_NONE, false);
    case 1: return SPR_LG(SPR_ACT_NONE, true);
    case 2: return SPR_LG(SPR_ACT_RELU, false);
    case 3: return SPR_LG(SPR_ACT_RELU, true);
    case 4: return SPR_LG(SPR_ACT_SIGMOID, false);
    default: return SPR_LG(SPR_ACT_SIGMOID, true);
  }
#undef SPR_LG
}

extern "C" int spr_linear(const float* x, int m, int k, const float* w, int n, const float* bias,
                          const float* residual, int act, float* out, void* ws, size_t ws_bytes,
                          void* stream_) {
  return spr_linear_r(x, m, k, w, n, bias, residual, act, out, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, ws, ws_bytes,
                      stream_);
}

extern "C" int spr_set_gemm_mode(int mode) {
  SPR_REQUIRE(mode == 0 || mode == 1, "gemm mode must be 0 (exact f32 MFMA) or 1 (split-fp16)");
  spr::g_gemm_mode.store(mode);
  return 0;
}

extern "C" int spr_layernorm_range_count(int m) {
  (void)m;
  return kLnRangeSlots;
}

extern "C" int spr_layernorm_r(const float* x, int m, int c, const float* gamma, const float* beta,
                               float eps, const float* pos, float* out_norm, float* out_pos, float* range_norm,
                               float* range_pos, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(m > 0 && c % 64 == 0 && c <= 1024, "layernorm: c must be a multiple of 64 and <= 1024 (c=%d)", c);
  SPR_REQUIRE(out_pos == nullptr || pos != nullptr, "layernorm: out_pos needs pos");
  const int grid = cdiv((long)m * 64, 256);
  if (c == 256)
    hipLaunchKernelGGL(k_layernorm256, dim3(cdiv((long)cdiv(m, kLnRows) * 64, 256)), dim3(256), 0, stream, x, m, gamma, beta, eps, pos, out_norm,
                       out_pos, range_norm, range_pos);
  else if (c <= 256)
    hipLaunchKernelGGL(k_layernorm<4>, dim3(grid), dim3(256), 0, stream, x, m, c, gamma, beta, eps,
                       pos, out_norm, out_pos, range_norm, range_pos);
  else
    hipLaunchKernelGGL(k_layernorm<16>, dim3(grid), dim3(256), 0, stream, x, m, c, gamma, beta, eps,
                       pos, out_norm, out_pos, range_norm, range_pos);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_layernorm(const float* x, int m, int c, const float* gamma, const float* beta,
                             float eps, const float* pos, float* out_norm, float* out_pos,
                             void* stream_) {
  return spr_layernorm_r(x, m, c, gamma, beta, eps, pos, out_norm, out_pos, nullptr, nullptr, stream_);
}

extern "C" int spr_posemb_sine(const float* xyz, int n, int d_model, float scale, float temperature,
                               float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n > 0 && d_model >= 6, "posemb: bad sizes");
  const int npf = d_model / 3 / 2 * 2;  // position_embedding.py:21
  const long total = (long)n * d_model;
  hipLaunchKernelGGL(k_posemb, dim3(cdiv(total, 256)), dim3(256), 0, stream, xyz, n, d_model, npf,
                     scale, temperature, out);
  SPR_LAUNCH_CHECK();
  return 0;
}
