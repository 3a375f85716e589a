// a4 -- KPConv forward (rigid kernel points, linear influence, sum
// aggregation) on gfx950.
//
// Behaviour contract: KPConv.forward
//   /root/reference/src/models/backbone_kpconv/kpconv_blocks.py:269-414
//     :309-315  shadow support point / centre neighbourhoods on the query
//     :325-329  squared distances to the kernel points
//     :368      w = clamp(1 - sqrt(d2)/KP_extent, min 0)
//     :388-394  weighted_features[n,p,:] = sum_k w[n,p,k] * x[idx[n,k],:]
//     :401-406  out[n,:] = sum_p weighted_features[n,p,:] @ W[p]
//     :409-412  divide by max(1, #{k : sum_c x[idx[n,k],c] > 0})
//
// The reference materialises [N,K,15,3] and [N,K,Cin] temporaries and runs
// two batched matmuls.  Here one fused kernel per call (k_kpconv_mfma, details
// at its definition):
//   phase 1 (per wave, per query): the influence x feature contraction is an
//     exact-f32 MFMA (v_mfma_f32_16x16x4_f32): lane (p = l&15, j = l>>4)
//     computes ONE influence weight w[p][neighbour 4s+j] -- exactly the
//     A-operand layout -- and loads a row slice of neighbour j as the B
//     operand; neighbour indices are staged in LDS, positions + the
//     neighbour-count flag come as one 16-byte record per support point;
//   phase 2 (per workgroup): the [TQ x 15*CC] weighted-feature tile (split
//     fp16 hi/lo in LDS) is contracted with W[15*Cin, Cout] by
//     v_mfma_f32_16x16x32_f16; W streams from L2 in MFMA-fragment order.
//   Channels are processed in chunks of CC <= 64 (one 123 KB LDS tile per
//   workgroup); phase-2 accumulators persist across chunks.
// Pre-kernels per call: k_rowflag (flag = sum_c x[i,c] > 0 and the packed
// support records), k_w_prep (weights -> fragment-order hi/lo planes).
// cin == 1 (the first block) has its own kernel, k_kpconv_cin1.
#include "spr_common.h"

namespace spr {
namespace {

constexpr int kKP = 15;  // kernel points handled by the MFMA path (padded to 16)

// Per support point: flag = (sum of its features > 0) (the reference's neighbour
// count, kpconv_blocks.py:399-404) and one 16-byte record {x, y, z, flag} so
// that the fused kernel fetches a neighbour's position and flag with ONE load.
__global__ __launch_bounds__(256) void k_rowflag(const float* __restrict__ x, const float* __restrict__ s_xyz,
                                                 int ns, int cin, unsigned char* __restrict__ flag,
                                                 float4* __restrict__ sxf) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int c4 = cin >> 2;
  if ((cin & 3) == 0 && c4 <= 64 && (c4 & (c4 - 1)) == 0) {
    // float4 per lane; a wave load covers 64 / c4 whole rows (1 KiB), kRfSteps loads in flight
    constexpr int kRfSteps = 4;
    const int rpl = 64 / c4;                       // rows per wave load
    const int sub = lane / c4, chunk = lane % c4;
    const int row0 = wid * rpl * kRfSteps + sub;
    float4 v[kRfSteps];
#pragma unroll
    for (int r = 0; r < kRfSteps; ++r) {
      const int row = row0 + r * rpl;
      v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < ns) v[r] = reinterpret_cast<const float4*>(x + (size_t)row * cin)[chunk];
    }
#pragma unroll
    for (int r = 0; r < kRfSteps; ++r) {
      const int row = row0 + r * rpl;
      // The flag is the SIGN of a float sum, so the summation order is part of the result (rows
      // of normalised features sum to ~0).  Keep the order of the one-wave-per-row form this
      // replaces (the generic path below): channel-index bits >= 6 ascending, then a butterfly
      // over bits 5..0 from the top -- here bits >= 2 live in the lane index (chunk), bits 1..0
      // inside the float4.
      float c[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = c[e];
        if (c4 == 64) {
          const int b = lane & 15;
          const float a0 = __shfl(t, b, 64), a1 = __shfl(t, b | 16, 64), a2 = __shfl(t, b | 32, 64),
                      a3 = __shfl(t, b | 48, 64);
          t = ((a0 + a1) + a2) + a3;
        } else if (c4 == 32) {
          t += __shfl_xor(t, 16, 64);
        }
        for (int o = (c4 < 16 ? c4 : 16) >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        c[e] = t;
      }
      const float t = (c[0] + c[2]) + (c[1] + c[3]);
      if (chunk == 0 && row < ns) {
        const int f = t > 0.f ? 1 : 0;
        flag[row] = (unsigned char)f;
        sxf[row] = make_float4(s_xyz[3 * (size_t)row], s_xyz[3 * (size_t)row + 1],
                               s_xyz[3 * (size_t)row + 2], __int_as_float(f));
      }
    }
    return;
  }
  // generic: one wave per row
  const int row = wid;
  if (row >= ns) return;
  float s = 0.f;
  for (int c = lane; c < cin; c += 64) s += x[(size_t)row * cin + c];
  s = wave_sum(s);
  if (lane == 0) {
    const int f = s > 0.f ? 1 : 0;
    flag[row] = (unsigned char)f;
    sxf[row] = make_float4(s_xyz[3 * (size_t)row], s_xyz[3 * (size_t)row + 1],
                           s_xyz[3 * (size_t)row + 2], __int_as_float(f));
  }
}

// ---------------------------------------------------------------------------
// Simple reference kernel (impl = 1, and the fallback for shapes the MFMA
// path does not cover): one thread per (query, output channel).
__global__ void k_kpconv_simple(const float* __restrict__ q_xyz, int nq,
                                const float* __restrict__ s_xyz, int ns,
                                const int* __restrict__ nbr, int nbr_stride, int kmax,
                                const float* __restrict__ x, int cin,
                                const float* __restrict__ W, int cout,
                                const float* __restrict__ kpts, int n_kp, float inv_extent,
                                const unsigned char* __restrict__ flag,
                                float* __restrict__ out) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)nq * cout) return;
  const int n = (int)(gid / cout), o = (int)(gid % cout);
  const float qx = q_xyz[3 * (size_t)n], qy = q_xyz[3 * (size_t)n + 1], qz = q_xyz[3 * (size_t)n + 2];
  float acc = 0.f;
  int cnt = 0;
  for (int k = 0; k < kmax; ++k) {
    const int idx = nbr[(size_t)n * nbr_stride + k];
    if (idx < 0 || idx >= ns) continue;
    cnt += flag[idx];
    const float rx = s_xyz[3 * (size_t)idx] - qx, ry = s_xyz[3 * (size_t)idx + 1] - qy,
                rz = s_xyz[3 * (size_t)idx + 2] - qz;
    for (int p = 0; p < n_kp; ++p) {
      const float dx = rx - kpts[3 * p], dy = ry - kpts[3 * p + 1], dz = rz - kpts[3 * p + 2];
      const float w = fmaxf(0.f, 1.f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent);
      if (w > 0.f) {
        float d = 0.f;
        for (int c = 0; c < cin; ++c)
          d += x[(size_t)idx * cin + c] * W[((size_t)p * cin + c) * cout + o];
        acc += w * d;
      }
    }
  }
  out[(size_t)n * cout + o] = acc / (float)max(cnt, 1);
}

// ---------------------------------------------------------------------------
// Cin == 1 (first block: features are a column of ones, qk_regtr_full.py:157).
// A wave owns 16 queries.  Lane (q = lane & 15, g = lane >> 4) accumulates the
// influence sums of kernel points g, g + 4, g + 8, g + 12 of query q -- which is
// exactly the A-operand layout of v_mfma_f32_16x16x4_f32 (row = query, k = g) --
// so the [16 q x 16 kp] x [16 kp x Cout] product that follows needs no
// transposition, and its C layout (lane = channel) stores 64-byte row pieces.
__global__ __launch_bounds__(256) void k_kpconv_cin1(
    const float* __restrict__ q_xyz, int nq, const float* __restrict__ s_xyz, int ns,
    const int* __restrict__ nbr, int nbr_stride, int kmax, int rows_sorted,
    const float* __restrict__ x, const float* __restrict__ W, int cout,
    const float* __restrict__ kpts, int n_kp, float inv_extent, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qv = lane & 15, g = lane >> 4;
  const int q0 = blockIdx.x * 64 + wave * 16;
  if (q0 >= nq) return;
  const int n = min(q0 + qv, nq - 1);
  const float qx = q_xyz[3 * (size_t)n], qy = q_xyz[3 * (size_t)n + 1], qz = q_xyz[3 * (size_t)n + 2];
  float kx[4], ky[4], kz[4];
  bool kok[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int p = 4 * s + g;
    kok[s] = p < n_kp;
    kx[s] = kok[s] ? kpts[3 * p] : 0.f;
    ky[s] = kok[s] ? kpts[3 * p + 1] : 0.f;
    kz[s] = kok[s] ? kpts[3 * p + 2] : 0.f;
  }
  float wf[4] = {0.f, 0.f, 0.f, 0.f};
  int cnt = 0;
  const int* row = nbr + (size_t)n * nbr_stride;
  for (int k = 0; k < kmax; ++k) {
    const int idx = row[k];
    const bool ok = idx >= 0 && idx < ns;
    if (rows_sorted && __ballot(ok) == 0ull) break;   // only trailing shadow entries left
    const size_t id = ok ? (size_t)idx : 0;
    const float xv = ok ? x[id] : 0.f;
    cnt += xv > 0.f ? 1 : 0;
    const float rx = s_xyz[3 * id] - qx, ry = s_xyz[3 * id + 1] - qy, rz = s_xyz[3 * id + 2] - qz;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float dx = rx - kx[s], dy = ry - ky[s], dz = rz - kz[s];
      // v_sqrt_f32 (1 ulp), like the fused kernel
      const float w = fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent);
      wf[s] += kok[s] ? w * xv : 0.f;
    }
  }
  const float inv = 1.f / (float)max(cnt, 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) wf[s] *= inv;
  for (int o0 = 0; o0 < cout; o0 += 16) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = 4 * s + g;
      const float bv = (p < n_kp && o0 + qv < cout) ? W[(size_t)p * cout + o0 + qv] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s], bv, acc, 0, 0, 0);
    }
    // C layout: row (query) = 4 g + r, col (channel) = qv
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qn = q0 + 4 * g + r;
      if (qn < nq && o0 + qv < cout) out[(size_t)qn * cout + o0 + qv] = acc[r];
    }
  }
}

// ---------------------------------------------------------------------------
// Weight preparation for phase 2 of the fused kernel: W [15*cin, cout] f32 ->
// hi / lo fp16 planes (w = hi + lo, lo = fp16(w - hi)) in FRAGMENT ORDER: the
// B operand of one v_mfma_f32_16x16x32_f16 (16 output channels x 32 k) is 64
// lanes x 8 halves = 1 KiB contiguous, so a wave's 16-byte-per-lane load
// covers eight full 128-byte lines instead of sixteen 64-byte pieces of
// sixteen different rows.
//   index = (((chunk * (cout/16) + nt) * KS + ks) * 64 + lane) * 8 + e
//   lane = (j4 << 4) | p16;  n = 16 nt + p16;  kk = 32 ks + 8 j4 + e (inside the
//   chunk's [15 x cc] block): kernel point kk / cc, channel chunk * cc + kk % cc
// The weights are multiplied by the power of two that brings max |w| into [2^14, 2^15)
// (w_parts: launch_absmax partials) before the split -- see split_pk_s, spr_common.h.
__global__ __launch_bounds__(256) void k_w_prep(const float* __restrict__ W, int cin, int cout, int cc,
                                                const float* __restrict__ w_parts, int n_wparts,
                                                _Float16* __restrict__ Wh, _Float16* __restrict__ Wl) {
  __shared__ float sh[17];
  const float sb = pow2f(pow2_exp_for(block_absmax(w_parts, sh, n_wparts)));
  const int total = kKP * cin * cout;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int KS = kKP * cc / 32, NT = cout / 16;
  const int e = i & 7, lane = (i >> 3) & 63;
  int r = i >> 9;
  const int ks = r % KS;
  r /= KS;
  const int nt = r % NT, chunk = r / NT;
  const int p16 = lane & 15, j4 = lane >> 4;
  const int kk = 32 * ks + 8 * j4 + e;
  const int p = kk / cc, c = chunk * cc + kk % cc;
  const float w = W[((size_t)p * cin + c) * cout + 16 * nt + p16] * sb;
  const _Float16 h = (_Float16)w;
  Wh[i] = h;
  Wl[i] = (_Float16)(w - (float)h);
}

__device__ __forceinline__ void wait_vm(int n) {   // n folds to a constant after unrolling
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

typedef _Float16 kh8 __attribute__((ext_vector_type(8)));
template <int N> struct VecF;
template <> struct VecF<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecF<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <int N> struct VecH;
template <> struct VecH<2> { typedef _Float16 type __attribute__((ext_vector_type(2))); };
template <> struct VecH<4> { typedef _Float16 type __attribute__((ext_vector_type(4))); };

// ---------------------------------------------------------------------------
// Fused MFMA kernel.
//   CC    channel chunk (32 or 64); NTC = CC/16 phase-1 n-tiles
//   TQ    queries per workgroup (multiple of 16); MT = TQ/16 m-tiles
//   NTW   phase-2 n-tiles (of 16 output channels) per wave
//   NW    phase-2 waves; SK = 1: wave -> (m-tile w % MT, n-group w / MT);
//         SK = 2: wave -> (n-group, k-half), every wave all m-tiles, the two
//         k-halves summed through LDS.  cout == 16 * NTW * (number of n-groups)
//   P1W   waves of the workgroup (phase 1 uses all of them; P1W >= NW)
// Wave w: phase 1 -> queries [w*TQ/P1W, (w+1)*TQ/P1W).
// Phase 2 streams the whole [15*Cin, Cout] weight matrix from L2 once per
// workgroup (in fragment order: whole 128-byte lines per wave load), so TQ is
// as large as LDS allows: 32 queries x 64 channels = 123 KB.
//
// Phase 1 (exact f32): "items" = (query, block of 16 neighbours); only LIVE
// blocks are enumerated (rows are distance sorted with trailing shadow
// entries, so a query with v valid neighbours owns ceil(v/16) items).  The
// item loop is software pipelined: while item i feeds the MFMAs, the gathers
// of item i+1 are in flight; neighbour indices come from LDS.  Lane (p, j)
// computes ONE influence weight per k-step = the A-operand layout of
// v_mfma_f32_16x16x4_f32; the B operand is one NTC-wide load per neighbour
// (lane p takes channels NTC*p .. NTC*p+NTC-1: 16 lanes = one whole row slice).
// Phase 2 (split fp16, fp32-level accuracy -- see linear.hip): the weighted
// features leave phase 1 as hi/lo fp16 tiles in LDS and are contracted with the
// pre-split transposed weights by v_mfma_f32_16x16x32_f16 (3 per 32-deep
// k-step); the weights stream from L2 through a ring of D k-steps issued by
// inline asm and retired with counted vmcnt.
template <int NTC>
struct KpItem {
  int idx[4];
  float4 sp[4];     // neighbour {x, y, z, flag bits}, one 16-byte load per k-step
  float b[4][NTC];
  float qx, qy, qz;
};

template <int CC, int TQ, int NTW, int NW, int SK, int P1W>
__global__ __launch_bounds__(64 * P1W) void k_kpconv_mfma(
    const float* __restrict__ q_xyz, int nq, const float* __restrict__ s_xyz, int ns,
    const int* __restrict__ nbr, int nbr_stride, int kmax, int rows_sorted,
    const float* __restrict__ x, int cin, const _Float16* __restrict__ Wh,
    const _Float16* __restrict__ Wl, int cout, const float* __restrict__ kpts, float inv_extent,
    const float4* __restrict__ sxf, const float* __restrict__ x_parts,
    const float* __restrict__ w_parts, int n_xparts, int n_wparts, float* __restrict__ out) {
  constexpr int NTC = CC / 16;
  constexpr int MT = TQ / 16;
  constexpr int KW = kKP * CC;       // phase-2 K per chunk
  constexpr int SH = KW + 16;        // LDS row stride in halves: conflict-free ds_read_b128
  constexpr int QPW = TQ / P1W;      // queries per wave in phase 1
  constexpr int NTHR = 64 * P1W;
  static_assert(P1W >= NW && TQ % P1W == 0, "phase-1 waves");
  extern __shared__ __align__(16) unsigned char lds_raw[];
  _Float16* wfh = (_Float16*)lds_raw;                 // [TQ][SH] hi
  _Float16* wfl = wfh + TQ * SH;                      // [TQ][SH] lo
  int* lcnt = (int*)(wfl + TQ * SH);                  // [TQ]
  int* lnit = lcnt + TQ;                              // [P1W] live items per wave
  int* litem = lnit + P1W;                            // [P1W][QPW * nblk] (qi << 8) | (last << 7) | b
  const int nblk = (kmax + 15) >> 4;                  // neighbour blocks per query
  const int KP = nblk * 16;
  int* lidx = litem + P1W * QPW * nblk;               // [TQ][KP] neighbour indices of the tile

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int p16 = lane & 15, j4 = lane >> 4;

  // kernel point of this lane (lane 15 of each 16 is padding)
  float kx = 0.f, ky = 0.f, kz = 0.f;
  if (p16 < kKP) {
    kx = kpts[3 * p16];
    ky = kpts[3 * p16 + 1];
    kz = kpts[3 * p16 + 2];
  }

  // Phase-2 operand scales (powers of two, split_pk_s): weighted features are bounded by
  // (valid neighbours) * max|x| <= kmax * max|x| (influences lie in [0, 1]); the weights were
  // scaled by k_w_prep from the same partials.
  float sa, unscale;
  {
    float* shf = reinterpret_cast<float*>(lds_raw);
    const int ka = pow2_exp_for(block_absmax(x_parts, shf, n_xparts) * (float)kmax);
    const int kb = pow2_exp_for(block_absmax(w_parts, shf, n_wparts));
    __syncthreads();
    sa = pow2f(ka);
    unscale = pow2f(-ka - kb);
  }

  // Persistent workgroup: tiles blockIdx.x, blockIdx.x + gridDim.x, ... (one LDS-sized
  // workgroup per CU, so a launch per tile only added dispatch latency between tiles).
  const int ntiles = (nq + TQ - 1) / TQ;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int q0 = tile * TQ;

  // Stage the tile's neighbour rows in LDS once (coalesced), padded with the
  // shadow index: the gather pipeline then depends on LDS reads only.
  for (int e = tid; e < TQ * KP; e += NTHR) {
    const int q = e / KP, k = e - q * KP;
    const int n = q0 + q;
    lidx[e] = (n < nq && k < kmax) ? nbr[(size_t)n * nbr_stride + k] : ns;
  }
  __syncthreads();
  // live-item list of this wave
  {
    int n_it = 0;
    for (int qi = 0; qi < QPW; ++qi) {
      const int* row = lidx + (wave * QPW + qi) * KP;
      int nb = nblk;
      if (rows_sorted) {   // valid count = position of the first shadow entry
        int v = 0;   // one ballot per 64 slots instead of a shuffle tree
        for (int k0 = 0; k0 < KP; k0 += 64) {
          const int k = k0 + lane;
          const bool ok = k < KP && row[min(k, KP - 1)] >= 0 && row[min(k, KP - 1)] < ns;
          v += __popcll(__ballot(ok));
        }
        nb = max(1, (v + 15) >> 4);
      }
      if (lane < nb) litem[wave * QPW * nblk + n_it + lane] = (qi << 8) | ((lane == nb - 1) ? 128 : 0) | lane;
      n_it += nb;
    }
    if (lane == 0) lnit[wave] = n_it;
  }
  __syncthreads();
  const int n_items = lnit[wave];
  const int* my_items = litem + wave * QPW * nblk;

  // phase-2 roles.  SK == 1: wave -> (m-tile wave % MT, n-group wave / MT), whole k range.
  // SK == 2: wave -> (n-group wave % NG, k-half wave / NG) and ALL m-tiles: every weight
  // fragment is fetched once per workgroup instead of once per m-tile (the weight stream
  // through the CU's vector memory path is what bounds this phase); the two k-halves are
  // summed through LDS at the end.
  constexpr int NG = (SK == 1) ? NW / MT : NW / SK;
  constexpr int MTW = (SK == 1) ? 1 : MT;                 // m-tiles per wave
  const int mt = (SK == 1) ? wave % MT : 0;
  const int ng = (SK == 1) ? wave / MT : wave % NG;
  const int kh = (SK == 1) ? 0 : wave / NG;
  f32x4 acc2[MTW][NTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc2[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < cin; c0 += CC) {
    // ------------------------------ phase 1 --------------------------------
    f32x4 acc1[NTC];
#pragma unroll
    for (int t = 0; t < NTC; ++t) acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cnt = 0;

    // issue the gathers of live item `it` (out-of-range items re-issue item 0: harmless)
    auto issue = [&](int it, KpItem<NTC>& I) {
      const int code = my_items[min(it, n_items - 1)];
      const int qi = code >> 8, b = code & 127;
      const int n = min(q0 + wave * QPW + qi, nq - 1);
      const int* row = lidx + (wave * QPW + qi) * KP + b * 16 + j4;
      I.qx = q_xyz[3 * (size_t)n];
      I.qy = q_xyz[3 * (size_t)n + 1];
      I.qz = q_xyz[3 * (size_t)n + 2];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int id_ = row[4 * s];
        I.idx[s] = id_;
        const bool ok = id_ >= 0 && id_ < ns;
        const size_t id = ok ? (size_t)id_ : 0;
        I.sp[s] = sxf[id];
        typedef typename VecF<NTC>::type vec_t;
        const vec_t xv = *reinterpret_cast<const vec_t*>(x + id * cin + c0 + NTC * p16);
#pragma unroll
        for (int t = 0; t < NTC; ++t) I.b[s][t] = xv[t];
      }
    };
    // consume item `it`; flush the query's accumulators after its last block
    auto compute = [&](int it, const KpItem<NTC>& I) {
      if (it >= n_items) return;
      const int code = my_items[it];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = I.idx[s] >= 0 && I.idx[s] < ns;
        const float dx = (I.sp[s].x - I.qx) - kx, dy = (I.sp[s].y - I.qy) - ky,
                    dz = (I.sp[s].z - I.qz) - kz;
        // v_sqrt_f32 (1 ulp) instead of the correctly-rounded expansion
        float w = fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent);
        if (!ok || p16 >= kKP) w = 0.f;
        cnt += (ok && c0 == 0 && p16 == 0) ? __float_as_int(I.sp[s].w) : 0;
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
          // a shadow slot gathered row 0 (finite) and has w = 0
          acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, I.b[s][t], acc1[t], 0, 0, 0);
        }
      }
      if (code & 128) {   // last live block of the query
        const int ql = wave * QPW + (code >> 8);
        // C/D layout: row (kernel point) = 4*j4 + r, col = p16 <-> channels NTC*p16 + t
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 4 * j4 + r;
          typedef typename VecH<NTC>::type hv_t;
          hv_t hh, ll;
          if constexpr (NTC == 2) {
            unsigned int hu, lu;
            split_pk_s(acc1[0][r], acc1[1][r], sa, hu, lu);
            hh = __builtin_bit_cast(hv_t, hu);
            ll = __builtin_bit_cast(hv_t, lu);
          } else {
            static_assert(NTC == 4, "NTC is 2 or 4");
            typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
            unsigned int h0, l0, h1, l1;
            split_pk_s(acc1[0][r], acc1[1][r], sa, h0, l0);
            split_pk_s(acc1[2][r], acc1[3][r], sa, h1, l1);
            hh = __builtin_bit_cast(hv_t, (u2_t){h0, h1});
            ll = __builtin_bit_cast(hv_t, (u2_t){l0, l1});
          }
          if (p < kKP) {
            *reinterpret_cast<hv_t*>(wfh + ql * SH + p * CC + NTC * p16) = hh;
            *reinterpret_cast<hv_t*>(wfl + ql * SH + p * CC + NTC * p16) = ll;
          }
        }
#pragma unroll
        for (int t = 0; t < NTC; ++t) acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c0 == 0) {
          int c = cnt;  // lanes with p16 == 0 hold the partial counts (one per j4)
          c += __shfl_xor(c, 16, 64);
          c += __shfl_xor(c, 32, 64);
          if (lane == 0) lcnt[ql] = c;
        }
        cnt = 0;
      }
    };

    {
      KpItem<NTC> A, B;
      issue(0, A);
      for (int it = 0; it < n_items; it += 2) {
        issue(it + 1, B);
        compute(it, A);
        issue(it + 2, A);
        compute(it + 1, B);
      }
    }
    __syncthreads();
    // ------------------------------ phase 2 --------------------------------
    if (wave < NW) {   // the first NW waves contract; extra phase-1 waves wait at the barrier
      constexpr int NBATCH = KW / 32 / SK;                          // 32-deep k-steps of this wave
      const int ks0 = kh * NBATCH;
      constexpr int LPB = 2 * NTW;                                  // 16-byte loads per k-step
      constexpr int D = (NTW <= 2) ? 5 : 3;                         // k-steps in flight
      static_assert(NBATCH % D == 0 && NBATCH >= 2 * D && (D - 1) * LPB <= 16, "ring shape");
      const _Float16* ah_row = wfh + (mt * 16 + p16) * SH + 8 * j4;
      const _Float16* al_row = wfl + (mt * 16 + p16) * SH + 8 * j4;
      constexpr int KS = KW / 32;                                  // k-steps per channel chunk
      // fragment-order planes (k_w_prep): 512 halves per (chunk, n-tile, k-step)
      const size_t wofs = ((size_t)(c0 / CC) * (cout / 16) + ng * NTW) * KS * 512 + lane * 8;
      kh8 bh[D][NTW], bl[D][NTW];
      auto load_step = [&](int ks, kh8 (&dh)[NTW], kh8 (&dl)[NTW]) {
        const size_t gk = wofs + (size_t)(ks0 + ks) * 512;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const _Float16* ph = Wh + gk + (size_t)t * KS * 512;
          const _Float16* pl = Wl + gk + (size_t)t * KS * 512;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dh[t]) : "v"(ph));
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dl[t]) : "v"(pl));
        }
      };
      auto mma_step = [&](int ks, const kh8 (&sh)[NTW], const kh8 (&sl)[NTW]) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
          const kh8 ah = *reinterpret_cast<const kh8*>(ah_row + m * 16 * SH + 32 * (ks0 + ks));
          const kh8 al = *reinterpret_cast<const kh8*>(al_row + m * 16 * SH + 32 * (ks0 + ks));
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            acc2[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, sl[t], acc2[m][t], 0, 0, 0);
            acc2[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, sh[t], acc2[m][t], 0, 0, 0);
            acc2[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, sh[t], acc2[m][t], 0, 0, 0);
          }
        }
      };
      wait_vm(0);
#pragma unroll
      for (int d = 0; d < D; ++d) load_step(d, bh[d], bl[d]);
      for (int b0 = 0; b0 < NBATCH - D; b0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          wait_vm((D - 1) * LPB);
          __builtin_amdgcn_sched_barrier(0);
          mma_step(b0 + d, bh[d], bl[d]);
          __builtin_amdgcn_sched_barrier(0);
          load_step(b0 + d + D, bh[d], bl[d]);
        }
      }
#pragma unroll
      for (int d = 0; d < D; ++d) {   // drain
        wait_vm((D - 1 - d) * LPB);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(NBATCH - D + d, bh[d], bl[d]);
      }
    }
    __syncthreads();
  }
  // ------------------------------ epilogue ---------------------------------
  const bool writer = (SK == 1) ? (wave < NW) : (wave < NW && kh == 0);
  if (SK == 2) {
    // sum the two k-halves through LDS (the wf tiles are dead after the last barrier)
    float* red = reinterpret_cast<float*>(lds_raw);            // [NG][MTW][NTW][4][64]
    if (wave < NW && kh == 1) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            red[(((ng * MTW + m) * NTW + t) * 4 + r) * 64 + lane] = acc2[m][t][r];
    }
    __syncthreads();
    if (writer) {
#pragma unroll
      for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc2[m][t][r] += red[(((ng * MTW + m) * NTW + t) * 4 + r) * 64 + lane];
    }
  }
  if (writer) {
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = (mt + m) * 16 + 4 * j4 + r;
        const int n = q0 + ql;
        if (n < nq) {
          const float inv = unscale / (float)max(lcnt[ql], 1);   // unscale: exact power of two
#pragma unroll
          for (int t = 0; t < NTW; ++t)
            out[(size_t)n * cout + (ng * NTW + t) * 16 + p16] = acc2[m][t][r] * inv;
        }
      }
  }
  __syncthreads();   // the next tile rewrites the LDS regions read above
  }  // persistent tile loop
}

template <int CC, int TQ, int NTW, int NW, int SK, int P1W = NW>
int launch_mfma(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                int nbr_stride, int kmax, int rows_sorted, const float* x, int cin,
                const _Float16* Wh, const _Float16* Wl, int cout, const float* kpts,
                float inv_extent, const float4* sxf, const float* x_parts, const float* w_parts, int n_xparts,
                int n_wparts, float* out, hipStream_t stream) {
  constexpr int SH = kKP * CC + 16;
  constexpr int QPW = TQ / P1W;
  const int nblk = (kmax + 15) / 16;
  const size_t lds = 2 * sizeof(_Float16) * (size_t)TQ * SH + sizeof(int) * (TQ + P1W) +
                     sizeof(int) * (size_t)P1W * QPW * nblk + sizeof(int) * (size_t)TQ * nblk * 16;
  auto kern = k_kpconv_mfma<CC, TQ, NTW, NW, SK, P1W>;
  ProfScope prof(stream, cin * 100000 + cout, nq);
  SPR_REQUIRE(lds <= 160 * 1024, "kpconv: neighbour rows too wide for the LDS tile (kmax=%d)", kmax);
  if (lds > 64 * 1024)
    if (int rc = ensure_dyn_lds((const void*)kern, 160 * 1024)) return rc;
  const int n_cu = device_cu_count();
  const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;   // LDS-limited residency
  const int ntiles = cdiv(nq, TQ);
  const int grid = ntiles < n_cu * per_cu ? ntiles : n_cu * per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * P1W), lds, stream, q_xyz, nq, s_xyz, ns,
                     nbr, nbr_stride, kmax, rows_sorted, x, cin, Wh, Wl, cout, kpts, inv_extent,
                     sxf, x_parts, w_parts, n_xparts, n_wparts, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_kpconv_workspace_bytes(int nq, int ns, int cin, int cout) {
  (void)nq;
  // flag bytes + {x,y,z,flag} support records + pre-split fragment-order weights (hi, lo fp16; up to
  // 32 kernel points)
  const size_t n = (size_t)(ns > 0 ? ns : 1);
  return align_up(n, 256) + align_up(16 * n, 256) + 2 * align_up((size_t)32 * cin * cout * 2, 256) +
         2 * align_up(kAmaxParts * sizeof(float), 256) + 256;
}

extern "C" int spr_kpconv_fwd(const float* q_xyz, int nq, const float* s_xyz, int ns,
                              const int* nbr, int nbr_stride, int kmax, int rows_sorted,
                              const float* x, int cin, const float* weights, int cout,
                              const float* kernel_points, int n_kp, float kp_extent,
                              float* out, int impl, void* ws, size_t ws_bytes,
                              void* stream_) {
  return spr_kpconv_fwd_r(q_xyz, nq, s_xyz, ns, nbr, nbr_stride, kmax, rows_sorted, x, cin, weights, cout, kernel_points,
                          n_kp, kp_extent, out, impl, nullptr, 0, nullptr, 0, ws, ws_bytes, stream_);
}

// x_range / w_range: operand ranges handed in (see spr_linear_r); NULL = measured here.
extern "C" int spr_kpconv_fwd_r(const float* q_xyz, int nq, const float* s_xyz, int ns,
                                const int* nbr, int nbr_stride, int kmax, int rows_sorted,
                                const float* x, int cin, const float* weights, int cout,
                                const float* kernel_points, int n_kp, float kp_extent,
                                float* out, int impl, const float* x_range, int x_range_n, const float* w_range,
                                int w_range_n, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE((x_range == nullptr || x_range_n >= 1) && (w_range == nullptr || w_range_n >= 1),
              "kpconv: a range needs a count");
  SPR_REQUIRE(nq > 0 && ns > 0, "kpconv: empty input");
  SPR_REQUIRE(kmax >= 1 && kmax <= nbr_stride, "kpconv: bad kmax=%d stride=%d", kmax, nbr_stride);
  SPR_REQUIRE(cin >= 1 && cout >= 1 && n_kp >= 1 && n_kp <= 32, "kpconv: bad dims");
  SPR_REQUIRE(kp_extent > 0.f, "kpconv: KP_extent must be > 0");
  SPR_REQUIRE(ws_bytes >= spr_kpconv_workspace_bytes(nq, ns, cin, cout), "kpconv: workspace too small");
  unsigned char* flag = (unsigned char*)ws;
  float4* sxf = (float4*)((char*)ws + align_up((size_t)ns, 256));
  _Float16* wh = (_Float16*)((char*)sxf + align_up((size_t)ns * 16, 256));
  _Float16* wl = (_Float16*)((char*)wh + align_up((size_t)32 * cin * cout * 2, 256));
  float* x_parts = (float*)((char*)wl + align_up((size_t)32 * cin * cout * 2, 256));
  float* w_parts = x_parts + align_up(kAmaxParts * sizeof(float), 256) / sizeof(float);
  const float inv_extent = 1.0f / kp_extent;

  if (cin == 1 && impl == 0 && n_kp <= 16) {
    ProfScope prof(stream, cin * 100000 + cout, nq);
    hipLaunchKernelGGL(k_kpconv_cin1, dim3(cdiv(nq, 64)), dim3(256), 0, stream, q_xyz, nq, s_xyz, ns, nbr,
                       nbr_stride, kmax, rows_sorted, x, weights, cout, kernel_points, n_kp, inv_extent, out);
    SPR_LAUNCH_CHECK();
    return 0;
  }

  {
    const int c4 = cin >> 2;
    const bool wide = (cin & 3) == 0 && c4 <= 64 && (c4 & (c4 - 1)) == 0;
    const long waves = wide ? cdiv(ns, (64 / c4) * 4) : (long)ns;     // matches k_rowflag's two layouts
    hipLaunchKernelGGL(k_rowflag, dim3(cdiv(waves * 64, 256)), dim3(256), 0, stream, x, s_xyz, ns, cin, flag, sxf);
  }
  SPR_LAUNCH_CHECK();

  if (impl == 0 && n_kp == kKP && cin % 32 == 0 && cout % 32 == 0 && cout <= 256) {
    const int ktot = n_kp * cin;
    const float* xp = x_range != nullptr ? x_range : x_parts;
    const float* wp = w_range != nullptr ? w_range : w_parts;
    const int n_xp = x_range != nullptr ? x_range_n : kAmaxParts, n_wp = w_range != nullptr ? w_range_n : kAmaxParts;
    if (x_range == nullptr && w_range == nullptr) {
      if (int rc = launch_absmax2(x, ns, cin, cin, x_parts, weights, ktot, cout, cout, w_parts, stream)) return rc;
    } else if (x_range == nullptr) {
      if (int rc = launch_absmax(x, ns, cin, cin, x_parts, stream)) return rc;
    } else if (w_range == nullptr) {
      if (int rc = launch_absmax(weights, ktot, cout, cout, w_parts, stream)) return rc;
    }
    hipLaunchKernelGGL(k_w_prep, dim3(cdiv((long)ktot * cout, 256)), dim3(256), 0, stream, weights, cin, cout,
                       cin % 64 == 0 ? 64 : 32, wp, n_wp, wh, wl);
#define SPR_KP_ARGS                                                                         \
  q_xyz, nq, s_xyz, ns, nbr, nbr_stride, kmax, rows_sorted, x, cin, wh, wl, cout,           \
      kernel_points, inv_extent, sxf, xp, wp, n_xp, n_wp, out, stream
    if (cin % 64 == 0) {
      // TQ = 32 (MT = 2), 8 waves: 4 n-groups x 2 k-halves, every wave both m-tiles
      if (cout == 64) return launch_mfma<64, 32, 1, 8, 2>(SPR_KP_ARGS);
      if (cout == 128) return launch_mfma<64, 32, 2, 8, 2>(SPR_KP_ARGS);
      if (cout == 256) return launch_mfma<64, 32, 4, 8, 2>(SPR_KP_ARGS);
    } else {
      // TQ = 64 (MT = 4), 8 waves: 2 n-groups
      if (cout == 32) return launch_mfma<32, 64, 1, 8, 1>(SPR_KP_ARGS);
      if (cout == 64) return launch_mfma<32, 64, 2, 8, 1>(SPR_KP_ARGS);
      if (cout == 128) return launch_mfma<32, 64, 4, 8, 1>(SPR_KP_ARGS);
    }
#undef SPR_KP_ARGS
  }
  // generic fallback / impl == 1
  const long total = (long)nq * cout;
  hipLaunchKernelGGL(k_kpconv_simple, dim3(cdiv(total, 256)), dim3(256), 0, stream, q_xyz, nq,
                     s_xyz, ns, nbr, nbr_stride, kmax, x, cin, weights, cout, kernel_points, n_kp,
                     inv_extent, flag, out);
  SPR_LAUNCH_CHECK();
  return 0;
}
