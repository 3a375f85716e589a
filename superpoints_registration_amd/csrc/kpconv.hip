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
// two batched matmuls.  Here one fused kernel per call:
//   phase 1 (per wave, per query): the influence x feature contraction is a
//     16x(16*t)x4 exact-f32 MFMA (v_mfma_f32_16x16x4_f32): lane (p = l&15,
//     j = l>>4) computes ONE influence weight w[p][neighbour 4s+j] -- that is
//     exactly the A-operand layout -- and loads 16 consecutive channels of
//     neighbour j as the B operand.  Neighbour rows are gathered straight
//     from HBM/L2 (64-byte runs per 16 lanes), 4 k-steps (16 neighbours) of
//     loads are issued before the first use.
//   phase 2 (per workgroup): the [TQ x 15*CC] weighted-feature tile is
//     transposed through LDS (row stride = 2 mod 32 words -> conflict free
//     A-fragment reads) and contracted with W[15*Cin, Cout] by the same MFMA
//     with queries on the M axis; W streams from L2.
//   Channels are processed in chunks of CC <= 64 so the LDS tile stays at
//   ~61 KB (2 workgroups per CU); phase-2 accumulators persist across chunks.
// The neighbour-count normaliser needs sum_c x[i,c] > 0 per support point:
// a 1-pass pre-kernel writes one flag byte per support point.
#include <vector>

#include "spr_common.h"

namespace spr {
namespace {

constexpr int kKP = 15;  // kernel points handled by the MFMA path (padded to 16)

__global__ void k_rowflag(const float* __restrict__ x, int ns, int cin,
                          unsigned char* __restrict__ flag) {
  // one wave per row
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= ns) return;
  float s = 0.f;
  for (int c = lane; c < cin; c += 64) s += x[(size_t)row * cin + c];
  s = wave_sum(s);
  if (lane == 0) flag[row] = s > 0.f ? 1 : 0;
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
// One thread per query accumulates the 15 influence sums; the 15 x Cout
// weight matrix sits in LDS.
template <int MAXKP>
__global__ __launch_bounds__(256) void k_kpconv_cin1(
    const float* __restrict__ q_xyz, int nq, const float* __restrict__ s_xyz, int ns,
    const int* __restrict__ nbr, int nbr_stride, int kmax, const float* __restrict__ x,
    const float* __restrict__ W, int cout, const float* __restrict__ kpts, int n_kp,
    float inv_extent, float* __restrict__ out) {
  extern __shared__ float lw[];  // [n_kp * cout] + [n_kp*3]
  float* lk = lw + n_kp * cout;
  for (int i = threadIdx.x; i < n_kp * cout; i += blockDim.x) lw[i] = W[i];
  for (int i = threadIdx.x; i < n_kp * 3; i += blockDim.x) lk[i] = kpts[i];
  __syncthreads();
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nq) return;
  const float qx = q_xyz[3 * (size_t)n], qy = q_xyz[3 * (size_t)n + 1], qz = q_xyz[3 * (size_t)n + 2];
  float wf[MAXKP];
#pragma unroll
  for (int p = 0; p < MAXKP; ++p) wf[p] = 0.f;
  int cnt = 0;
  for (int k = 0; k < kmax; ++k) {
    const int idx = nbr[(size_t)n * nbr_stride + k];
    if (idx < 0 || idx >= ns) continue;
    const float xv = x[idx];
    cnt += xv > 0.f ? 1 : 0;
    const float rx = s_xyz[3 * (size_t)idx] - qx, ry = s_xyz[3 * (size_t)idx + 1] - qy,
                rz = s_xyz[3 * (size_t)idx + 2] - qz;
#pragma unroll
    for (int p = 0; p < MAXKP; ++p) {
      if (p < n_kp) {
        const float dx = rx - lk[3 * p], dy = ry - lk[3 * p + 1], dz = rz - lk[3 * p + 2];
        const float w = fmaxf(0.f, 1.f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent);
        wf[p] += w * xv;
      }
    }
  }
  const float inv = 1.f / (float)max(cnt, 1);
  for (int o = 0; o < cout; ++o) {
    float a = 0.f;
#pragma unroll
    for (int p = 0; p < MAXKP; ++p)
      if (p < n_kp) a += wf[p] * lw[p * cout + o];
    out[(size_t)n * cout + o] = a * inv;
  }
}

// ---------------------------------------------------------------------------
// Fused MFMA kernel.
//   CC    channel chunk (16, 32 or 64); NTC = CC/16 phase-1 n-tiles
//   TQ    queries per workgroup (multiple of 16); MT = TQ/16 m-tiles
//   NTW   phase-2 n-tiles (of 16 output channels) per wave
// 256 threads = 4 waves.  Wave w: phase 1 -> queries [w*TQ/4, (w+1)*TQ/4);
// phase 2 -> m-tile (w % MT), n-tiles [(w / MT) * NTW, +NTW).
// Requires cout == 16 * NTW * (4 / MT).
template <int CC, int TQ, int NTW>
__global__ __launch_bounds__(256) void k_kpconv_mfma(
    const float* __restrict__ q_xyz, int nq, const float* __restrict__ s_xyz, int ns,
    const int* __restrict__ nbr, int nbr_stride, int kmax, int rows_sorted,
    const float* __restrict__ x, int cin, const float* __restrict__ W, int cout,
    const float* __restrict__ kpts, float inv_extent,
    const unsigned char* __restrict__ flag, float* __restrict__ out) {
  constexpr int NTC = CC / 16;
  constexpr int MT = TQ / 16;
  constexpr int KW = kKP * CC;       // phase-2 K per chunk
  constexpr int STRIDE = KW + 2;     // KW % 32 == 0 -> stride = 2 mod 32 words
  constexpr int QPW = TQ / 4;        // queries per wave in phase 1
  extern __shared__ __align__(16) float lds[];
  float* wf = lds;                   // [TQ][STRIDE]
  int* lcnt = (int*)(lds + TQ * STRIDE);  // [TQ]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int p16 = lane & 15, j4 = lane >> 4;
  const int q0 = blockIdx.x * TQ;

  // kernel point of this lane (lane 15 of each 16 is padding)
  float kx = 0.f, ky = 0.f, kz = 0.f;
  if (p16 < kKP) {
    kx = kpts[3 * p16];
    ky = kpts[3 * p16 + 1];
    kz = kpts[3 * p16 + 2];
  }

  const int mt = wave % MT;
  const int ng = wave / MT;
  f32x4 acc2[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < cin; c0 += CC) {
    // ------------------------------ phase 1 --------------------------------
    for (int qi = 0; qi < QPW; ++qi) {
      const int ql = wave * QPW + qi;
      const int n = q0 + ql;
      f32x4 acc1[NTC];
#pragma unroll
      for (int t = 0; t < NTC; ++t) acc1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int cnt = 0;
      if (n < nq) {  // wave-uniform
        const float qx = q_xyz[3 * (size_t)n], qy = q_xyz[3 * (size_t)n + 1],
                    qz = q_xyz[3 * (size_t)n + 2];
        const int* row = nbr + (size_t)n * nbr_stride;
        // blocks of 4 k-steps = 16 neighbours: issue all loads, then compute
        for (int kb = 0; kb < kmax; kb += 16) {
          int idx[4];
          bool ok[4];
          float sx[4], sy[4], sz[4];
          float b[4][NTC];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int k = kb + 4 * s + j4;
            idx[s] = (k < kmax) ? row[k] : ns;
            ok[s] = idx[s] >= 0 && idx[s] < ns;
          }
          if (rows_sorted) {
            const bool any = ok[0];  // first neighbour of the block, lanes j4==0
            if (__ballot(any) == 0ull) break;
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const size_t id = ok[s] ? (size_t)idx[s] : 0;
            sx[s] = s_xyz[3 * id];
            sy[s] = s_xyz[3 * id + 1];
            sz[s] = s_xyz[3 * id + 2];
#pragma unroll
            for (int t = 0; t < NTC; ++t) b[s][t] = x[id * cin + c0 + t * 16 + p16];
            if (c0 == 0 && p16 == 0 && ok[s]) cnt += flag[id];
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float dx = (sx[s] - qx) - kx, dy = (sy[s] - qy) - ky, dz = (sz[s] - qz) - kz;
            float w = fmaxf(0.f, 1.f - sqrtf(dx * dx + dy * dy + dz * dz) * inv_extent);
            if (!ok[s] || p16 >= kKP) w = 0.f;
#pragma unroll
            for (int t = 0; t < NTC; ++t) {
              const float bv = ok[s] ? b[s][t] : 0.f;
              acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, bv, acc1[t], 0, 0, 0);
            }
          }
        }
      }
      // C/D layout: row (kernel point) = 4*j4 + r, col (channel) = p16
#pragma unroll
      for (int t = 0; t < NTC; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 4 * j4 + r;
          if (p < kKP) wf[ql * STRIDE + p * CC + t * 16 + p16] = acc1[t][r];
        }
      if (c0 == 0) {
        // lanes with p16 == 0 hold partial counts (one per j4)
        int c = cnt;
        c += __shfl_xor(c, 16, 64);
        c += __shfl_xor(c, 32, 64);
        if (lane == 0) lcnt[ql] = c;
      }
    }
    __syncthreads();
    // ------------------------------ phase 2 --------------------------------
    {
      const float* arow = wf + (mt * 16 + p16) * STRIDE + j4;
      // W row for phase-2 k index kk (within chunk): p = kk / CC, c = kk % CC
      for (int k0 = 0; k0 < KW; k0 += 4) {
        const float a = arow[k0];
        const int kk = k0 + j4;
        const int p = kk / CC, c = kk % CC;
        const float* wrow = W + ((size_t)p * cin + c0 + c) * cout + (ng * NTW) * 16 + p16;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const float bv = wrow[t * 16];
          acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc2[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ------------------------------ epilogue ---------------------------------
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ql = mt * 16 + 4 * j4 + r;
    const int n = q0 + ql;
    if (n < nq) {
      const float inv = 1.f / (float)max(lcnt[ql], 1);
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        out[(size_t)n * cout + (ng * NTW + t) * 16 + p16] = acc2[t][r] * inv;
    }
  }
}

// ---- optional per-launch HIP-event timing (bench.py roofline leg) ------------
struct ProfRec {
  hipEvent_t beg, end;
  int code;  // cin * 100000 + cout
  int nq;
};
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;

struct ProfScope {
  hipStream_t stream;
  bool on;
  ProfRec rec;
  ProfScope(hipStream_t s, int cin, int cout, int nq) : stream(s), on(g_prof_on) {
    if (!on) return;
    rec.code = cin * 100000 + cout;
    rec.nq = nq;
    if (hipEventCreate(&rec.beg) != hipSuccess || hipEventCreate(&rec.end) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(rec.beg, stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(rec.end, stream);
    g_prof.push_back(rec);
  }
};

template <int CC, int TQ, int NTW>
int launch_mfma(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                int nbr_stride, int kmax, int rows_sorted, const float* x, int cin,
                const float* W, int cout, const float* kpts, float inv_extent,
                const unsigned char* flag, float* out, hipStream_t stream) {
  constexpr int STRIDE = kKP * CC + 2;
  const size_t lds = sizeof(float) * (size_t)TQ * STRIDE + sizeof(int) * TQ;
  auto kern = k_kpconv_mfma<CC, TQ, NTW>;
  ProfScope prof(stream, cin, cout, nq);
  if (lds > 64 * 1024) {
    SPR_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  }
  hipLaunchKernelGGL(kern, dim3(cdiv(nq, TQ)), dim3(256), lds, stream, q_xyz, nq, s_xyz, ns,
                     nbr, nbr_stride, kmax, rows_sorted, x, cin, W, cout, kpts, inv_extent,
                     flag, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_kpconv_workspace_bytes(int nq, int ns, int cin, int cout) {
  (void)nq;
  (void)cin;
  (void)cout;
  return align_up((size_t)(ns > 0 ? ns : 1), 256) + 256;
}

extern "C" int spr_kpconv_fwd(const float* q_xyz, int nq, const float* s_xyz, int ns,
                              const int* nbr, int nbr_stride, int kmax, int rows_sorted,
                              const float* x, int cin, const float* weights, int cout,
                              const float* kernel_points, int n_kp, float kp_extent,
                              float* out, int impl, void* ws, size_t ws_bytes,
                              void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(nq > 0 && ns > 0, "kpconv: empty input");
  SPR_REQUIRE(kmax >= 1 && kmax <= nbr_stride, "kpconv: bad kmax=%d stride=%d", kmax, nbr_stride);
  SPR_REQUIRE(cin >= 1 && cout >= 1 && n_kp >= 1 && n_kp <= 32, "kpconv: bad dims");
  SPR_REQUIRE(kp_extent > 0.f, "kpconv: KP_extent must be > 0");
  SPR_REQUIRE(ws_bytes >= spr_kpconv_workspace_bytes(nq, ns, cin, cout), "kpconv: workspace too small");
  unsigned char* flag = (unsigned char*)ws;
  const float inv_extent = 1.0f / kp_extent;

  if (cin == 1 && impl == 0 && n_kp <= 16) {
    const size_t lds = sizeof(float) * ((size_t)n_kp * cout + 3 * n_kp);
    hipLaunchKernelGGL(k_kpconv_cin1<16>, dim3(cdiv(nq, 256)), dim3(256), lds, stream, q_xyz, nq,
                       s_xyz, ns, nbr, nbr_stride, kmax, x, weights, cout, kernel_points, n_kp,
                       inv_extent, out);
    SPR_LAUNCH_CHECK();
    return 0;
  }

  hipLaunchKernelGGL(k_rowflag, dim3(cdiv((long)ns * 64, 256)), dim3(256), 0, stream, x, ns, cin,
                     flag);
  SPR_LAUNCH_CHECK();

  if (impl == 0 && n_kp == kKP && cin % 16 == 0) {
#define SPR_KP_ARGS                                                                         \
  q_xyz, nq, s_xyz, ns, nbr, nbr_stride, kmax, rows_sorted, x, cin, weights, cout,          \
      kernel_points, inv_extent, flag, out, stream
    if (cin % 64 == 0) {
      // TQ = 16 (MT = 1): the 4 waves split Cout
      if (cout == 64) return launch_mfma<64, 16, 1>(SPR_KP_ARGS);
      if (cout == 128) return launch_mfma<64, 16, 2>(SPR_KP_ARGS);
      if (cout == 256) return launch_mfma<64, 16, 4>(SPR_KP_ARGS);
      if (cout == 512) return launch_mfma<64, 16, 8>(SPR_KP_ARGS);
    } else if (cin % 32 == 0) {
      // TQ = 32 (MT = 2): 2 n-groups
      if (cout == 32) return launch_mfma<32, 32, 1>(SPR_KP_ARGS);
      if (cout == 64) return launch_mfma<32, 32, 2>(SPR_KP_ARGS);
      if (cout == 128) return launch_mfma<32, 32, 4>(SPR_KP_ARGS);
    } else {
      // CC = 16, TQ = 64 (MT = 4): each wave all n-tiles
      if (cout == 16) return launch_mfma<16, 64, 1>(SPR_KP_ARGS);
      if (cout == 32) return launch_mfma<16, 64, 2>(SPR_KP_ARGS);
      if (cout == 64) return launch_mfma<16, 64, 4>(SPR_KP_ARGS);
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

// ---- profiling control (see include/spr.h) -----------------------------------
extern "C" int spr_prof_enable(int on) {
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  g_prof.clear();
  g_prof_on = on != 0;
  return 0;
}

extern "C" int spr_prof_read(int max_records, int* codes, int* nqs, float* ms) {
  int n = 0;
  for (auto& r : g_prof) {
    if (n >= max_records) break;
    if (hipEventSynchronize(r.end) != hipSuccess) break;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.beg, r.end) != hipSuccess) break;
    codes[n] = r.code;
    nqs[n] = r.nq;
    ms[n] = t;
    ++n;
  }
  return n;
}
