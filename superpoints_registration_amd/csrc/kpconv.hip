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
                                                 float4* __restrict__ sxf, int* __restrict__ tile_ctr) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int c4 = cin >> 2;
  // record ns = the shadow support point of the ring kernel: far away from everything (influence 0
  // on every kernel point), flag 0
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sxf[ns] = make_float4(-1.0e17f, -1.0e17f, -1.0e17f, __int_as_float(0));
    tile_ctr[0] = 0;   // the ring kernel's tile hand-out counter (same stream, next launch)
  }
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

// Support records for the Cin == 1 kernel: {x, y, z, feature} in one 16-byte load per neighbour
// instead of three coordinate loads and a feature load.
__global__ __launch_bounds__(256) void k_records_cin1(const float* __restrict__ x, const float* __restrict__ s_xyz,
                                                      int ns, float4* __restrict__ sxf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) sxf[i] = make_float4(s_xyz[3 * (size_t)i], s_xyz[3 * (size_t)i + 1], s_xyz[3 * (size_t)i + 2], x[i]);
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
    const float4* __restrict__ sxf, const float* __restrict__ W, int cout,
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
    const float4 rec = sxf[id];          // {x, y, z, feature}: one 16-byte load (the four kernel-point groups of a query share it)
    const float xv = ok ? rec.w : 0.f;
    cnt += xv > 0.f ? 1 : 0;
    const float rx = rec.x - qx, ry = rec.y - qy, rz = rec.z - qz;
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


// ===========================================================================
// Ring kernel (round 3): the same contraction with the neighbour gather taken
// OFF the critical path.
//
// What the round-2 kernel lost: its
// gathers were register loads, so a wave had at most two 16-neighbour items in
// flight, each item walked the chain LDS index -> gather -> influence -> MFMA
// serially, the [15 Cin, Cout] weights were re-streamed from L2 for every 32
// queries (as many bytes as the gather itself) and nothing moved while the
// tile's index rows were staged.  scripts/abl/kp_gather.hip measured what the
// memory side can do on the bench's own index matrices: 43-45 GB/s per CU
// (11 TB/s chip-wide) for 256-byte rows fetched by LDS-DMA, already with 16-32
// KiB in flight per CU -- three to four times what that kernel drew.
//
// Structure (one persistent 8-wave workgroup per CU, tiles of TQ = 16 queries):
//   * a tile descriptor table (k_kp_tiles, cached per neighbour matrix) gives
//     wave w of tile t its two queries -- the 16 queries sorted by live item
//     count and dealt serpentine (rank w and rank 15-w) so that the eight waves
//     reach the tile barrier together -- and their item counts;
//   * an ITEM is (query, block of 8 neighbours).  Its 8 feature rows and 8
//     {x,y,z,flag} records arrive by global_load_lds_dwordx4 (per-lane source
//     address = a row gather) in a per-wave LDS ring of NS items, issued NS-1
//     items ahead and retired with COUNTED s_waitcnt vmcnt: no VGPR staging, the
//     loads of the next tile's first items fly during phase 2 and the barriers.
//     The wave's two index rows for tile t+1 come the same way, two tiles ahead;
//   * phase 1 per item: 2 k-steps of v_mfma_f32_16x16x4_f32 exactly as before
//     (lane (kernel point, neighbour) computes one influence = the A layout; the B
//     operand is a ds_read of the gathered row), shadow neighbours carry a
//     far-away record (influence 0) instead of a select;
//   * phase 2: the [16 x 15 Cin] weighted-feature tile (split fp16 in LDS) times W,
//     with W held in REGISTERS for the whole launch (15 Cin Cout fp16 hi/lo =
//     61-245 KB over 8 waves x <=120 VGPRs: wave = (16-channel group, k-slice)):
//     nothing is streamed, so a 16-query tile costs no more weight traffic than a
//     64-query one; the k-slices are summed through LDS in a fixed order.
//   Two raw s_barriers per tile (wf complete / wf free); outputs are bitwise
//   independent of the tile walk and of the ring depth.
// Covers Cin in {32, 64} with Cin * Cout <= 4096 (the 32->32 and 64->64 layers:
// every KPConv of levels 0 and 1); other shapes keep k_kpconv_mfma.
constexpr int kRingTQ = 16;
#ifndef SPR_KP_NS64
#define SPR_KP_NS64 4   // ring depth of the 64 -> 64 instantiation (experiment builds override it)
#endif

// Tile descriptors: one wave per tile of 16 queries (in `order` if given).  Entry [tile][w] =
// {query A, query B, items A, items B} for wave w (-1 / 0 beyond nq).  An item = 8 neighbour
// slots; rows_sorted: live items = ceil(valid / 8) (>= 1), else every slot block.
__global__ __launch_bounds__(256) void k_kp_tiles(const int* __restrict__ nbr, int nq, int ns, int nbr_stride,
                                                  int kmax, int rows_sorted, const int* __restrict__ order,
                                                  int ntiles, int4* __restrict__ desc, int* __restrict__ pad_word) {
  const int tile = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) pad_word[0] = ns;   // what padded slots of a staged index row read
  if (tile >= ntiles) return;
  int my_q = -1, my_c = 0;
  for (int r = 0; r < kRingTQ; ++r) {
    const int pos = tile * kRingTQ + r;
    int q = -1, c = 0;
    if (pos < nq) {
      q = order ? order[pos] : pos;
      if (rows_sorted) {
        int v = 0;
        for (int k0 = 0; k0 < kmax; k0 += 64) {
          const int k = k0 + lane;
          const int id = k < kmax ? nbr[(size_t)q * nbr_stride + k] : ns;
          v += __popcll(__ballot(id >= 0 && id < ns));
        }
        c = max(1, (v + 7) >> 3);
      } else {
        c = (kmax + 7) >> 3;
      }
    }
    if (lane == r) { my_q = q; my_c = c; }
  }
  // rank by (items descending, position ascending) among the 16 entries
  int rank = 0;
  for (int j = 0; j < kRingTQ; ++j) {
    const int cj = __shfl(my_c, j, 64);
    rank += (cj > my_c || (cj == my_c && j < lane)) ? 1 : 0;
  }
  if (lane < kRingTQ) {
    int* d = reinterpret_cast<int*>(desc + (size_t)tile * 8);
    const int w = rank < 8 ? rank : 15 - rank, second = rank < 8 ? 0 : 1;
    d[w * 4 + second] = my_q;
    d[w * 4 + 2 + second] = my_c;
  }
}

// LDS-DMA: 16 bytes per active lane from base + off (per-lane byte offset) to LDS at dst_s + 16 * lane.
// The wait in front orders the write behind every ds_read this wave has issued (the slot being refilled).
__device__ __forceinline__ void ring_dma16(const void* base, unsigned off, unsigned dst_s) {
  unsigned keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(off), "s"(base), "s"(dst_s) : "memory");
}
__device__ __forceinline__ void ring_dma16_nw(const void* base, unsigned off, unsigned dst_s) {   // no LDS wait
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(off), "s"(base), "s"(dst_s) : "memory");
}
// m0 is not preserved (declared as a clobber: the compiler keeps nothing in it across the statement)
__device__ __forceinline__ void ring_dma16_m0(const void* base, unsigned off, unsigned dst_s) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(off), "s"(base), "s"(dst_s) : "memory", "m0");
}
// lanes 0..7 only (8 records of 16 bytes -> dst_s .. dst_s + 128); exec is restored inside the statement
__device__ __forceinline__ void ring_dma16_rec8(const void* base, unsigned off, unsigned dst_s) {
  unsigned long long keep;
  asm volatile(
      "s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xff\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
      : "=&s"(keep) : "v"(off), "s"(base), "s"(dst_s) : "memory", "m0");
}
// All DMA instructions of one ring item behind ONE M0 setup (round 5): the instruction's immediate offset moves the
// global AND the LDS address, so the second row piece (LDS slot + 1 KiB) and the record piece (slot + 8 RB) are issued
// against bases moved DOWN by their LDS displacement.  PPI = row pieces per item (1: 32 channels, 2: 64 channels).
template <int PPI, int RB>
__device__ __forceinline__ void ring_item_dma(const void* x, const void* x_m1k, const void* sx_m, unsigned off0,
                                              unsigned off1, unsigned off_rec, unsigned dst_s) {
  unsigned long long keep;
  if constexpr (PPI == 2) {
    asm volatile(
        "s_mov_b32 m0, %7\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
        "s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xff\n\t"
        "global_load_lds_dwordx4 %3, %6 offset:%8\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(keep) : "v"(off0), "v"(off1), "v"(off_rec), "s"(x), "s"(x_m1k), "s"(sx_m), "s"(dst_s), "n"(8 * RB)
        : "memory", "m0");
  } else {
    asm volatile(
        "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xff\n\t"
        "global_load_lds_dwordx4 %2, %4 offset:%6\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(keep) : "v"(off0), "v"(off_rec), "s"(x), "s"(sx_m), "s"(dst_s), "n"(8 * RB)
        : "memory", "m0");
  }
}
__device__ __forceinline__ void ring_dma4(const void* base, unsigned off, unsigned dst_s) {
  unsigned keep;
  asm volatile(
      "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(off), "s"(base), "s"(dst_s) : "memory");
}
__device__ __forceinline__ void wait_vm_any(int n) {   // n is wave uniform
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
    case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
    case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int CC, int COUT, int NS>
struct RingShape {
  static constexpr int NTC = CC / 16;                 // phase-1 n-tiles
  static constexpr int KW = kKP * CC;                 // phase-2 K
  static constexpr int SH = KW + 16;                  // wf row stride (halves): conflict-free ds_read_b128
  static constexpr int KS = KW / 32;                  // 32-deep k-steps
  static constexpr int NG = COUT / 16;                // 16-channel output groups
  static constexpr int SPLIT = 8 / NG;                // k-slices (waves per group)
  static constexpr int KSW = (KS + SPLIT - 1) / SPLIT;  // k-steps per wave (the last slice may hold fewer)
  static constexpr int RB = CC * 4;                   // feature row bytes
  static constexpr int LPR = RB / 16;                 // lanes per row in a 1-KiB piece
  static constexpr int RPP = 64 / LPR;                // rows per piece
  static constexpr int PPI = 8 / RPP;                 // row pieces per item
  static constexpr int DPI = PPI + 1;                 // DMA instructions per item (+ the record piece)
  static constexpr int SLOT = 8 * RB + 128;           // ring slot bytes: 8 rows + 8 records
  static constexpr int WF_BYTES = 2 * kRingTQ * SH * 2;
  static constexpr int RED_BYTES = NG * (SPLIT - 1) * 4 * 64 * 4;
  static constexpr int RING_BYTES = 8 * NS * SLOT;
  static constexpr int SMALL_BYTES = 2 * kRingTQ * 4 * 2 + 16;   // lcnt[2][16], lqid[2][16], ltile[4]
  // 32 -> 32 fits 128 VGPRs and (with NS = 3) 80 KB of LDS: two workgroups = 16 waves per CU
  static constexpr int MINW = (CC == 32 && COUT == 32) ? 4 : 2;   // waves per SIMD the register budget is cut for
  static_assert(NG * SPLIT == 8 && NS >= 3, "shape");
  static __host__ __device__ size_t lds_bytes(int idxw) {                 // idxw: ints per staged index row (64 or 128)
    return (size_t)WF_BYTES + RED_BYTES + RING_BYTES + SMALL_BYTES + (size_t)8 * 2 * 2 * idxw * 4;
  }
};

#ifdef SPR_KP_RING_PROF
// Experiment builds only (make EXTRA=-DSPR_KP_RING_PROF): workgroup 0 records a shader-clock time line
// of its first tiles -- per wave up to kTraceN stamps {id, s_memtime} in LDS, dumped to g_kp_trace at
// the end.  ids: 1 phase-1 start, 2 item start, 3 item loop done, 4 primed, 5 past barrier 1,
// 6 phase 2 done, 7 past barrier 2, 8 epilogue done.
constexpr int kTraceN = 160;
__device__ unsigned long long g_kp_trace[8 * kTraceN];
#define KP_STAMP(id)                                                                            \
  do {                                                                                          \
    if (blockIdx.x == 0 && trace_n < kTraceN) {                                                 \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                               \
      if (lane == 0) trace_lds[wave * kTraceN + trace_n] = (t_ << 4) | (unsigned long long)(id); \
      ++trace_n;                                                                                \
    }                                                                                           \
  } while (0)
#else
#define KP_STAMP(id)
#endif

template <int CC, int COUT, int NS>
__global__ __launch_bounds__(512, (RingShape<CC, COUT, NS>::MINW)) void k_kpconv_ring(
    const float* __restrict__ q_xyz, int ns, const int* __restrict__ nbr, int nbr_stride, int kmax,
    const float* __restrict__ x, const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl,
    const float* __restrict__ kpts, float inv_extent, const float4* __restrict__ sxf,
    const int4* __restrict__ desc, int ntiles, const int* __restrict__ pad_word, int idxw,
    const float* __restrict__ x_parts, const float* __restrict__ w_parts, int n_xparts, int n_wparts,
    int* __restrict__ tile_ctr, float* __restrict__ out) {
  typedef RingShape<CC, COUT, NS> S;
  constexpr int NTC = S::NTC, SH = S::SH, KS = S::KS, NG = S::NG, SPLIT = S::SPLIT, KSW = S::KSW;
  constexpr int RB = S::RB, LPR = S::LPR, RPP = S::RPP, PPI = S::PPI, DPI = S::DPI, SLOT = S::SLOT;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  _Float16* wfh = (_Float16*)lds_raw;                                   // [16][SH] hi
  _Float16* wfl = wfh + kRingTQ * SH;                                   // [16][SH] lo
  float* red = (float*)(lds_raw + S::WF_BYTES);                         // [NG][SPLIT-1][4][64]
  unsigned char* ring_all = lds_raw + S::WF_BYTES + S::RED_BYTES;       // [8][NS][SLOT]
  int* lcnt = (int*)(ring_all + S::RING_BYTES);                         // [2][16]
  int* lqid = lcnt + 2 * kRingTQ;                                       // [2][16]
  int* ltile = lqid + 2 * kRingTQ;                                      // [4] tile hand-out (below)
  int* idx_all = ltile + 4;                                             // [8][2 parity][2 queries][idxw]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, j4 = lane >> 4;
  unsigned char* ring = ring_all + wave * (NS * SLOT);
  int* idxbuf = idx_all + wave * (4 * idxw);
  const unsigned ring_s = (unsigned)(uintptr_t)ring;
  const unsigned idx_s = (unsigned)(uintptr_t)idxbuf;

  // kernel point of this lane; lane 15 of each 16 is padding: parked far away -> influence 0
  float kx = 1.0e18f, ky = 1.0e18f, kz = 1.0e18f;
  if (p16 < kKP) {
    kx = kpts[3 * p16];
    ky = kpts[3 * p16 + 1];
    kz = kpts[3 * p16 + 2];
  }
  float sa, unscale;
  {
    float* shf = reinterpret_cast<float*>(lds_raw);
    const int ka = pow2_exp_for(block_absmax(x_parts, shf, n_xparts) * (float)kmax);
    const int kb = pow2_exp_for(block_absmax(w_parts, shf, n_wparts));
    __syncthreads();
    sa = pow2f(ka);
    unscale = pow2f(-ka - kb);
  }

  // phase-2 role and the wave's slice of W, resident in registers for the whole launch
  const int ng = wave % NG, sp = wave / NG;
  const int ks0 = sp * KSW;
  kh8 bh[KSW], bl[KSW];
#pragma unroll
  for (int k = 0; k < KSW; ++k) {
    const int ks = ks0 + k;
    if (ks < KS) {
      const size_t o = ((size_t)ng * KS + ks) * 512 + lane * 8;
      bh[k] = *reinterpret_cast<const kh8*>(Wh + o);
      bl[k] = *reinterpret_cast<const kh8*>(Wl + o);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) { bh[k][e] = (_Float16)0.f; bl[k][e] = (_Float16)0.f; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // everything the compiler issued so far has landed

  // ---- per-wave item stream ------------------------------------------------------------------
  // Index rows of a tile, staged by LDS-DMA into idxbuf[par]: query A's row at int 0, query B's row
  // right behind A's LIVE blocks (int 8 * items(A)), so that item j of the wave's stream finds its 8
  // neighbour ids at int 8 j whichever query it belongs to.  Lanes beyond kmax read a word holding
  // ns (= shadow).  A's DMA is issued first and lands first (loads return in order).
  const int idxb = 2 * idxw;                                   // ints per (wave, parity)
  auto stage_idx = [&](const int4 d, int par) {
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) {
      const int q = sel ? d.y : d.x;
      for (int h = 0; h < idxw; h += 64) {
        const int k = h + lane;
        const unsigned dst = __builtin_amdgcn_readfirstlane(idx_s + (par * idxb + (sel ? 8 * d.z : 0) + h) * 4);
        // two masked instructions, each with a uniform base; both write dst + 4 * lane
        if (q >= 0 && k < kmax) ring_dma4(nbr, (unsigned)(((size_t)q * nbr_stride + k) * 4), dst);
        else ring_dma4(pad_word, 0u, dst);
      }
    }
  };
  // The gathers of an item in two steps, so that the LDS reads of the neighbour ids are never on the
  // issue path: load_ids(item) one stage ahead, issue_ids(slot, ids) when the slot is free.
  struct Ids { unsigned row[PPI]; unsigned rec; };
  const unsigned a_idr = (lane / LPR) * 4, a_idc = (lane & 7) * 4;      // per-lane byte offsets inside an item's ids
  auto load_ids = [&](int par, int j) -> Ids {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(idxbuf) + (par * idxb + 8 * j) * 4;
    Ids r;
#pragma unroll
    for (int pc = 0; pc < PPI; ++pc) r.row[pc] = *reinterpret_cast<const unsigned*>(base + a_idr + pc * RPP * 4);
    r.rec = *reinterpret_cast<const unsigned*>(base + a_idc);
    return r;
  };
  // No LDS wait in front of these DMAs: a slot is refilled only after the item that lived in it has
  // been moved to registers and its first influence computed, and the LDS retires a wave's reads in
  // order, hundreds of cycles before a DMA can return.
  const unsigned a_dma = (lane % LPR) * 16;
  auto issue_ids = [&](unsigned slot_off, const Ids& ids) {
#ifdef SPR_KP_ABL_NODMA
    return;
#endif
    const unsigned slot = __builtin_amdgcn_readfirstlane(ring_s + slot_off);
#ifndef SPR_KP_RING_NO_LGKM
    // Every ds_read of the slot's previous item has RETURNED before the refill is issued (the
    // ordering used to rest on timing alone -- the LDS retires a wave's reads in order, hundreds of cycles before
    // a DMA can land -- which no ISA rule guarantees).  A/B against -DSPR_KP_RING_NO_LGKM: scripts/kp_lgkm_ab.sh.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#ifdef SPR_KP_ITEM_DMA_OLD
#pragma unroll
    for (int pc = 0; pc < PPI; ++pc) {
      const unsigned rid = min(ids.row[pc], (unsigned)(ns - 1));   // shadow slots fetch a real (finite) row
      ring_dma16_m0(x, rid * RB + a_dma, slot + pc * 1024);
    }
    ring_dma16_rec8(sxf, min(ids.rec, (unsigned)ns) * 16, slot + 8 * RB);   // record ns = the shadow record
#else
    static_assert(PPI == 1 || PPI == 2, "ring item: one or two row pieces");
    const unsigned o0 = min(ids.row[0], (unsigned)(ns - 1)) * RB + a_dma;      // shadow slots fetch a real (finite) row
    const unsigned o1 = min(ids.row[PPI - 1], (unsigned)(ns - 1)) * RB + a_dma;
    const unsigned orec = min(ids.rec, (unsigned)ns) * 16;                       // record ns = the shadow record
    ring_item_dma<PPI, RB>(x, reinterpret_cast<const char*>(x) - 1024, reinterpret_cast<const char*>(sxf) - 8 * RB, o0, o1,
                           orec, slot);
#endif
  };
  // first min(NS, n) items of a tile: all ids first, then all gathers
  auto prime = [&](const int4 d, int par) {
    const int n = d.z + d.w;
    Ids ids[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) ids[i] = load_ids(par, i);
#pragma unroll
    for (int i = 0; i < NS; ++i)
      if (i < n) issue_ids(i * SLOT, ids[i]);
  };

#ifdef SPR_KP_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses every issue arbitration otherwise
#endif
  // Tile walk: tiles are handed out by one device-wide counter (zeroed by k_rowflag ahead of this
  // launch) instead of blockIdx + k * gridDim.  The kernel shares the chip with the pyramid builder and
  // the shortcut branches of the same forward (other HIP streams): a workgroup that becomes resident
  // late -- its CU was busy -- then simply takes fewer tiles, where a static partition made the whole
  // launch wait for it.  A workgroup holds three tile indices at any time (current, next: gathers
  // primed, after-next: index rows staged); wave 0 draws the following one during phase 1 and
  // publishes it through LDS across barrier 2.  Outputs do not depend on the walk.
  if (tid == 0) {
    const int t0 = __hip_atomic_fetch_add(tile_ctr, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ltile[0] = t0;
  }
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(ltile[0]);
  int tile_nxt = tile + 1, tile_nn = tile + 2;
  __syncthreads();
  int4 d_cur = make_int4(-1, -1, 0, 0), d_nxt = make_int4(-1, -1, 0, 0);
  if (tile < ntiles) d_cur = desc[(size_t)tile * 8 + wave];
  if (tile_nxt < ntiles) d_nxt = desc[(size_t)tile_nxt * 8 + wave];
  if (tile < ntiles) {
    stage_idx(d_cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tile_nxt < ntiles) stage_idx(d_nxt, 1);
    prime(d_cur, 0);
  }

#ifdef SPR_KP_RING_PROF
  unsigned long long* trace_lds = reinterpret_cast<unsigned long long*>(lds_raw + S::lds_bytes(idxw));
  int trace_n = 0;
#endif
  typedef typename VecF<NTC>::type vec_t;
  struct KStep { float4 rec; vec_t xv; };
  // operands of k-step s2 (neighbours 4 s2 + j4) of the item in the slot at byte offset slot_off
  const unsigned a_rec = 8 * RB + j4 * 16, a_xv = j4 * RB + p16 * (NTC * 4);
  auto ld_kstep = [&](unsigned slot_off, int s2) -> KStep {
    KStep k;
    k.rec = *reinterpret_cast<const float4*>(ring + slot_off + a_rec + s2 * 64);
    k.xv = *reinterpret_cast<const vec_t*>(ring + slot_off + a_xv + s2 * 4 * RB);
    return k;
  };
  // Output rows of the previous tile, held back by the writer waves: vmcnt counts stores in issue
  // order with the gathers, so a store issued right after phase 2 would sit in FRONT of the next
  // tile's counted waits and expose its write latency there.  They leave at the end of the next
  // phase 1 instead, when the wave has just drained its queue anyway.
  float pend_v[4] = {0.f, 0.f, 0.f, 0.f};
  int pend_n[4] = {-1, -1, -1, -1};
  auto flush_pending = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (pend_n[r] >= 0) out[(size_t)pend_n[r] * COUT + ng * 16 + p16] = pend_v[r];
  };
  int par = 0;
  for (; tile < ntiles; par ^= 1) {
    const int4 d = d_cur;
    KP_STAMP(1);
    const int n_items = d.z + d.w;
    // descriptor two tiles ahead (scalar load; its index rows are staged at the end of this phase 1)
    int4 d_nn = make_int4(-1, -1, 0, 0);
    const bool has_nxt = tile_nxt < ntiles, has_nn = tile_nn < ntiles;
    if (has_nn) d_nn = desc[(size_t)tile_nn * 8 + wave];
    // the tile after those: drawn now, used after barrier 2 (the value returns during the item loop)
    int drawn = ntiles;
    if (tid == 0 && has_nn) drawn = __hip_atomic_fetch_add(tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ------------------------------ phase 1 --------------------------------
    // coordinates of the wave's two queries (scalar loads, issued here so that none sits in the item loop)
    float qax = 0.f, qay = 0.f, qaz = 0.f, qbx = 0.f, qby = 0.f, qbz = 0.f;
    if (d.x >= 0) { qax = q_xyz[3 * (size_t)d.x]; qay = q_xyz[3 * (size_t)d.x + 1]; qaz = q_xyz[3 * (size_t)d.x + 2]; }
    if (d.y >= 0) { qbx = q_xyz[3 * (size_t)d.y]; qby = q_xyz[3 * (size_t)d.y + 1]; qbz = q_xyz[3 * (size_t)d.y + 2]; }
    // Pipeline at k-step granularity with FIXED register roles: R0 / R1 hold the operands of k-step
    // 0 / 1 of the current item; each is reloaded from the next item's slot right after its MFMAs
    // have been issued, so an LDS read has a whole k-step of work in front of its first use, no
    // register set is ever copied, and the gathers of items i + 2 .. i + NS - 1 stay in flight.
    auto wait_item = [&](int i) {   // item i has landed: at most the younger items may be outstanding
      const int younger = n_items - 1 - i;
      if (younger >= NS - 2) wait_vm_any(DPI * (NS - 2));
      else wait_vm_any(DPI * max(younger, 0));
    };
    KStep R0, R1;
    if (n_items == 0) {
      wait_vm_any(0);
    } else {
      wait_item(0);
      R0 = ld_kstep(0, 0);
      R1 = ld_kstep(0, 1);
    }
    Ids ids_next = load_ids(par, NS);
    int i = 0;
    unsigned slot_off = 0;
    // the query's weighted features -> wf row ql (split fp16), its neighbour count and id
    auto flush = [&](const f32x4 (&acc)[NTC], int cnt, int ql, int qid) {
      // C/D layout: row (kernel point) = 4*j4 + r, col = p16 <-> channels NTC*p16 + t
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 4 * j4 + r;
        typedef typename VecH<NTC>::type hv_t;
        hv_t hh, ll;
        if constexpr (NTC == 2) {
          unsigned int hu, lu;
          split_pk_s(acc[0][r], acc[1][r], sa, hu, lu);
          hh = __builtin_bit_cast(hv_t, hu);
          ll = __builtin_bit_cast(hv_t, lu);
        } else {
          typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
          unsigned int h0, l0, h1, l1;
          split_pk_s(acc[0][r], acc[1][r], sa, h0, l0);
          split_pk_s(acc[2][r], acc[3][r], sa, h1, l1);
          hh = __builtin_bit_cast(hv_t, (u2_t){h0, h1});
          ll = __builtin_bit_cast(hv_t, (u2_t){l0, l1});
        }
        if (p < kKP) {
          *reinterpret_cast<hv_t*>(wfh + ql * SH + p * CC + NTC * p16) = hh;
          *reinterpret_cast<hv_t*>(wfl + ql * SH + p * CC + NTC * p16) = ll;
        }
      }
      int c = cnt;   // every lane of a 16-group saw the same records: one count per j4
      c += __shfl_xor(c, 16, 64);
      c += __shfl_xor(c, 32, 64);
      if (lane == 0) {
        lcnt[par * kRingTQ + ql] = c;
        lqid[par * kRingTQ + ql] = qid;
      }
    };
    // One item = two k-steps.  `between` runs after the first k-step's MFMAs have been issued: the
    // place where the previous query's flush goes, so that its conversions and LDS writes execute
    // under the matrix pipe's work instead of behind its drain.
    auto item = [&](f32x4 (&acc)[NTC], int& cnt, float qx, float qy, float qz, auto&& between) {
      KP_STAMP(2);
      const unsigned nslot = slot_off + SLOT == NS * SLOT ? 0u : slot_off + SLOT;
      {   // k-step 0
        const float dx = (R0.rec.x - qx) - kx, dy = (R0.rec.y - qy) - ky, dz = (R0.rec.z - qz) - kz;
        const float w = fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent);
        cnt += __float_as_int(R0.rec.w);
#ifndef SPR_KP_ABL_NOMFMA1
#pragma unroll
        for (int t = 0; t < NTC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, R0.xv[t], acc[t], 0, 0, 0);
#else
#pragma unroll
        for (int t = 0; t < NTC; ++t) acc[t][0] += w * R0.xv[t];
#endif
      }
      wait_item(i + 1);                   // (the last item of the tile: everything has landed)
      R0 = ld_kstep(nslot, 0);
      if (i + NS < n_items) issue_ids(slot_off, ids_next);   // refill the slot item i lived in
      between();
      {   // k-step 1
        const float dx = (R1.rec.x - qx) - kx, dy = (R1.rec.y - qy) - ky, dz = (R1.rec.z - qz) - kz;
        const float w = fmaxf(0.f, 1.f - __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent);
        cnt += __float_as_int(R1.rec.w);
#ifndef SPR_KP_ABL_NOMFMA1
#pragma unroll
        for (int t = 0; t < NTC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, R1.xv[t], acc[t], 0, 0, 0);
#else
#pragma unroll
        for (int t = 0; t < NTC; ++t) acc[t][0] += w * R1.xv[t];
#endif
      }
      R1 = ld_kstep(nslot, 1);
      ids_next = load_ids(par, i + NS + 1);   // (past the tile's last item: unused ints of the buffer)
      slot_off = nslot;
      ++i;
    };
    auto nothing = [] {};
    f32x4 accA[NTC], accB[NTC];
#pragma unroll
    for (int t = 0; t < NTC; ++t) accA[t] = accB[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int cntA = 0, cntB = 0;
#pragma unroll 1
    for (int b = 0; b < d.z; ++b) item(accA, cntA, qax, qay, qaz, nothing);
    if (d.w > 0) {
      // query B's first item carries query A's flush (query A exists whenever B does)
      item(accB, cntB, qbx, qby, qbz, [&] { flush(accA, cntA, wave, d.x); });
#pragma unroll 1
      for (int b = 1; b < d.w; ++b) item(accB, cntB, qbx, qby, qbz, nothing);
      flush(accB, cntB, 15 - wave, d.y);
    } else if (d.z > 0) {
      flush(accA, cntA, wave, d.x);
    }
    KP_STAMP(3);
    // rows of this tile that hold no query (tail tile): mark them so that the epilogue skips them
    if (lane == 0) {
      if (d.x < 0) lqid[par * kRingTQ + wave] = -1;
      if (d.y < 0) lqid[par * kRingTQ + 15 - wave] = -1;
    }
    // everything issued so far has landed (the last item waited for vmcnt(0)): the index rows of the
    // next tile are in LDS.  Stage those of the tile after it, then prime the ring with the next tile's
    // first items: they fly during phase 2, the barriers and the epilogue.
    if (sp == 0) flush_pending();
    if (tid == 0) ltile[1 + par] = drawn;
    if (has_nn) stage_idx(d_nn, par);
    if (has_nxt) prime(d_nxt, par ^ 1);
    KP_STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // B1: the wf tile is complete
    KP_STAMP(5);

    // ------------------------------ phase 2 --------------------------------
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    {
      const _Float16* ah_row = wfh + p16 * SH + 8 * j4;
      const _Float16* al_row = wfl + p16 * SH + 8 * j4;
#ifdef SPR_KP_ABL_NOP2
      if (ntiles < 0)
#endif
#pragma unroll
      for (int k = 0; k < KSW; ++k) {
        const int ks = min(ks0 + k, KS - 1);            // a slice short of KSW steps meets zero weights
        const kh8 ah = *reinterpret_cast<const kh8*>(ah_row + 32 * ks);
        const kh8 al = *reinterpret_cast<const kh8*>(al_row + 32 * ks);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[k], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[k], acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[k], acc2, 0, 0, 0);
      }
    }
    if (sp > 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((ng * (SPLIT - 1) + sp - 1) * 4 + r) * 64 + lane] = acc2[r];
    }
    KP_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // B2: wf free again, partial sums visible
    KP_STAMP(7);
    if (sp == 0) {
#pragma unroll
      for (int s2 = 1; s2 < SPLIT; ++s2)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[r] += red[((ng * (SPLIT - 1) + s2 - 1) * 4 + r) * 64 + lane];
      // C layout: row (query of the tile) = 4*j4 + r, col = p16
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 4 * j4 + r;
        pend_n[r] = lqid[par * kRingTQ + ql];
        const float inv = unscale / (float)max(lcnt[par * kRingTQ + ql], 1);   // unscale: exact power of two
        pend_v[r] = acc2[r] * inv;
      }
    }
    KP_STAMP(8);
    d_cur = d_nxt;
    d_nxt = d_nn;
    tile = tile_nxt;
    tile_nxt = tile_nn;
    tile_nn = __builtin_amdgcn_readfirstlane(ltile[1 + par]);   // published before barrier 1 of this tile
  }
  if (sp == 0) flush_pending();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SPR_KP_RING_PROF
  if (blockIdx.x == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int k = lane; k < kTraceN; k += 64) g_kp_trace[wave * kTraceN + k] = k < trace_n ? trace_lds[wave * kTraceN + k] : 0ull;
  }
#endif
}

template <int CC, int COUT, int NS>
int launch_ring(const float* q_xyz, int nq, int ns, const int* nbr, int nbr_stride, int kmax, const float* x,
                const _Float16* Wh, const _Float16* Wl, const float* kpts, float inv_extent, const float4* sxf,
                const int4* desc, const int* pad_word, const float* x_parts, const float* w_parts, int n_xparts,
                int n_wparts, int* tile_ctr, float* out, hipStream_t stream) {
  typedef RingShape<CC, COUT, NS> S;
  const int idxw = kmax <= 64 ? 64 : 128;
#ifdef SPR_KP_RING_PROF
  const size_t lds = S::lds_bytes(idxw) + 8 * kTraceN * 8;
#else
  const size_t lds = S::lds_bytes(idxw);
#endif
  auto kern = k_kpconv_ring<CC, COUT, NS>;
  ProfScope prof(stream, CC * 100000 + COUT, nq);
  if (lds > 64 * 1024)
    if (int rc = ensure_dyn_lds((const void*)kern, 160 * 1024)) return rc;
  const int ntiles = cdiv(nq, kRingTQ);
  const int n_cu = device_cu_count();
  int per_cu = (int)((160 * 1024) / lds);                       // LDS-limited residency
  per_cu = per_cu < 1 ? 1 : (per_cu > S::MINW / 2 ? S::MINW / 2 : per_cu);
  const int grid = ntiles < n_cu * per_cu ? ntiles : n_cu * per_cu;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, q_xyz, ns, nbr, nbr_stride, kmax, x, Wh, Wl, kpts,
                     inv_extent, sxf, desc, ntiles, pad_word, idxw, x_parts, w_parts, n_xparts, n_wparts, tile_ctr, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

}  // namespace
}  // namespace spr

using namespace spr;

#ifdef SPR_KP_RING_PROF
extern "C" int spr_debug_kp_trace(unsigned long long* out, int n) {   // n <= 8 * kTraceN
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kp_trace), sizeof(unsigned long long) * n) != hipSuccess) return 1;
  return 0;
}
#endif

static inline size_t plan_desc_bytes(int nq) { return align_up((size_t)cdiv(nq > 0 ? nq : 1, kRingTQ) * 8 * sizeof(int4), 256); }
static inline int w_chunk(int cin) { return cin % 64 == 0 ? 64 : 32; }
static inline bool ring_shape(int cin, int cout) {
  return (cin == 32 || cin == 64) && cin * cout <= 4096 && cout % 32 == 0 && 8 % (cout / 16) == 0;
}

extern "C" size_t spr_kpconv_plan_bytes(int nq) { return plan_desc_bytes(nq) + 256; }

extern "C" int spr_kpconv_plan(const int* nbr, int nq, int ns, int nbr_stride, int kmax, int rows_sorted,
                               const int* order, void* plan, size_t plan_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(nq > 0 && ns > 0 && kmax >= 1 && kmax <= nbr_stride, "kpconv plan: bad sizes");
  SPR_REQUIRE(plan != nullptr && plan_bytes >= spr_kpconv_plan_bytes(nq), "kpconv plan: buffer too small");
  const int ntiles = cdiv(nq, kRingTQ);
  int4* desc = (int4*)plan;
  int* pad_word = (int*)((char*)plan + plan_desc_bytes(nq));
  hipLaunchKernelGGL(k_kp_tiles, dim3(cdiv((long)ntiles * 64, 256)), dim3(256), 0, stream, nbr, nq, ns, nbr_stride,
                     kmax, rows_sorted, order, ntiles, desc, pad_word);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spr_kpconv_wplanes_bytes(int cin, int cout) { return 2 * align_up((size_t)32 * cin * cout * 2, 256); }

extern "C" int spr_kpconv_prep_weights(const float* weights, int n_kp, int cin, int cout, const float* w_range,
                                       int w_range_n, void* wplanes, size_t bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n_kp == kKP && cin % 32 == 0 && cout % 32 == 0 && cout <= 256,
              "kpconv weight planes exist for the MFMA shapes only (15 kernel points, channels in multiples of 32)");
  SPR_REQUIRE(w_range != nullptr && w_range_n >= 1, "kpconv weight planes: the weight range is required");
  SPR_REQUIRE(wplanes != nullptr && bytes >= spr_kpconv_wplanes_bytes(cin, cout), "kpconv weight planes: buffer too small");
  _Float16* wh = (_Float16*)wplanes;
  _Float16* wl = (_Float16*)((char*)wplanes + align_up((size_t)32 * cin * cout * 2, 256));
  hipLaunchKernelGGL(k_w_prep, dim3(cdiv((long)n_kp * cin * cout, 256)), dim3(256), 0, stream, weights, cin, cout,
                     w_chunk(cin), w_range, w_range_n, wh, wl);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spr_kpconv_workspace_bytes(int nq, int ns, int cin, int cout) {
  // flag bytes + {x,y,z,flag} support records + pre-split fragment-order weights (hi, lo fp16; up to
  // 32 kernel points)
  const size_t n = (size_t)(ns > 0 ? ns : 1);
  return align_up(n, 256) + align_up(16 * (n + 1), 256) + spr_kpconv_wplanes_bytes(cin, cout) +
         2 * align_up(kAmaxParts * sizeof(float), 256) + spr_kpconv_plan_bytes(nq) + 256;
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
  return spr_kpconv_fwd_p(q_xyz, nq, s_xyz, ns, nbr, nbr_stride, kmax, rows_sorted, x, cin, weights, cout, kernel_points,
                          n_kp, kp_extent, out, impl, x_range, x_range_n, w_range, w_range_n, nullptr, nullptr, ws,
                          ws_bytes, stream_);
}

// plan / wplanes: spr_kpconv_plan of THIS neighbour matrix (same nq, ns, stride, kmax, rows_sorted) and
// spr_kpconv_prep_weights of THESE weights with THIS w_range; NULL = built here, per call.
extern "C" int spr_kpconv_fwd_p(const float* q_xyz, int nq, const float* s_xyz, int ns,
                                const int* nbr, int nbr_stride, int kmax, int rows_sorted,
                                const float* x, int cin, const float* weights, int cout,
                                const float* kernel_points, int n_kp, float kp_extent,
                                float* out, int impl, const float* x_range, int x_range_n, const float* w_range,
                                int w_range_n, const void* plan, const void* wplanes, void* ws, size_t ws_bytes,
                                void* stream_) {
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
  _Float16* wh = (_Float16*)((char*)sxf + align_up(((size_t)ns + 1) * 16, 256));
  _Float16* wl = (_Float16*)((char*)wh + align_up((size_t)32 * cin * cout * 2, 256));
  float* x_parts = (float*)((char*)wl + align_up((size_t)32 * cin * cout * 2, 256));
  float* w_parts = x_parts + align_up(kAmaxParts * sizeof(float), 256) / sizeof(float);
  void* ws_plan = (char*)w_parts + align_up(kAmaxParts * sizeof(float), 256);
  int* tile_ctr = (int*)((char*)ws_plan + spr_kpconv_plan_bytes(nq));
  SPR_REQUIRE(wplanes == nullptr || w_range != nullptr, "kpconv: prepared weight planes come with the range they were scaled by");
  if (wplanes != nullptr) {
    wh = (_Float16*)wplanes;
    wl = (_Float16*)((char*)wplanes + align_up((size_t)32 * cin * cout * 2, 256));
  }
  const float inv_extent = 1.0f / kp_extent;

  if (cin == 1 && (impl == 0 || impl == 2) && n_kp <= 16) {
    hipLaunchKernelGGL(k_records_cin1, dim3(cdiv(ns, 256)), dim3(256), 0, stream, x, s_xyz, ns, sxf);
    ProfScope prof(stream, cin * 100000 + cout, nq);
    hipLaunchKernelGGL(k_kpconv_cin1, dim3(cdiv(nq, 64)), dim3(256), 0, stream, q_xyz, nq, s_xyz, ns, nbr,
                       nbr_stride, kmax, rows_sorted, sxf, weights, cout, kernel_points, n_kp, inv_extent, out);
    SPR_LAUNCH_CHECK();
    return 0;
  }

  {
    const int c4 = cin >> 2;
    const bool wide = (cin & 3) == 0 && c4 <= 64 && (c4 & (c4 - 1)) == 0;
    const long waves = wide ? cdiv(ns, (64 / c4) * 4) : (long)ns;     // matches k_rowflag's two layouts
    hipLaunchKernelGGL(k_rowflag, dim3(cdiv(waves * 64, 256)), dim3(256), 0, stream, x, s_xyz, ns, cin, flag, sxf, tile_ctr);
  }
  SPR_LAUNCH_CHECK();

  if ((impl == 0 || impl == 2) && n_kp == kKP && cin % 32 == 0 && cout % 32 == 0 && cout <= 256) {
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
    if (wplanes == nullptr)
      hipLaunchKernelGGL(k_w_prep, dim3(cdiv((long)ktot * cout, 256)), dim3(256), 0, stream, weights, cin, cout,
                         w_chunk(cin), wp, n_wp, wh, wl);
    // ring kernel: 32- / 64-channel inputs whose whole weight matrix fits the register file
    const bool ring_ok = impl == 0 && ring_shape(cin, cout) && kmax <= 128 && (size_t)ns * cin * 4 < (1ull << 32) &&
                         (size_t)nq * nbr_stride * 4 < (1ull << 32);
    if (ring_ok) {
      if (plan == nullptr) {
        if (int rc = spr_kpconv_plan(nbr, nq, ns, nbr_stride, kmax, rows_sorted, nullptr, ws_plan,
                                     spr_kpconv_plan_bytes(nq), stream_)) return rc;
        plan = ws_plan;
      }
      const int4* desc = (const int4*)plan;
      const int* pad_word = (const int*)((const char*)plan + plan_desc_bytes(nq));
#define SPR_RING_ARGS                                                                                       \
  q_xyz, nq, ns, nbr, nbr_stride, kmax, x, wh, wl, kernel_points, inv_extent, sxf, desc, pad_word, xp, wp, \
      n_xp, n_wp, tile_ctr, out, stream
      if (cin == 64 && cout == 64) return launch_ring<64, 64, SPR_KP_NS64>(SPR_RING_ARGS);
      if (cin == 64 && cout == 32) return launch_ring<64, 32, 4>(SPR_RING_ARGS);
      if (cin == 32 && cout == 32) return launch_ring<32, 32, 3>(SPR_RING_ARGS);
      if (cin == 32 && cout == 64) return launch_ring<32, 64, 8>(SPR_RING_ARGS);
      if (cin == 32 && cout == 128) return launch_ring<32, 128, 8>(SPR_RING_ARGS);
#undef SPR_RING_ARGS
    }
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
