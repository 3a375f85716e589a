// Layout of the split-fp16 operand planes the attention kernel consumes
// (attention.hip) and the in-projection GEMM can emit directly (linear.hip).
#pragma once
#include "spr_common.h"

namespace spr {

// Token columns of segment s in the transposed V planes start at a 32-byte
// aligned column; gaps and the tail hold zeros.
__host__ __device__ inline int attn_vstart_of(int cu_s, int s) { return (cu_s + 16 * s) & ~15; }
// Round 5: inside every aligned group of 16 columns the tokens are stored in the order [0-3, 8-11, 4-7, 12-15].
// The B fragment of O^T = V^T P^T holds, per k-step, the keys 16 s + 8 (j >> 2) + 4 h + (j & 3) (the 32x32
// accumulator layout of the scores): with this order they are 16 CONTIGUOUS bytes of a plane row, so the attention
// core stages V^T tiles by LDS-DMA (16-byte pieces cannot be permuted on the way) and reads a fragment with one
// ds_read_b128.  attn_vperm(column) = where a token column lives.
__host__ __device__ inline int attn_vperm(int col) { return (col & ~15) | (col & 3) | ((col & 4) << 1) | ((col & 8) >> 1); }
// Round 5: the "transposed" V planes are stored BLOCKED, [16-column group][feature][16 columns]: element (feature f,
// plane column c) of a plane with d features lives at ((c >> 4) d + f) 16 + (c & 15).  One head's 32 features of one
// 16-column group are 1 KiB contiguous -- the unit a wave of the fused chains writes with ONE coalesced store
// instruction (lane = (feature, 8 of the 16 columns), 16 bytes each; with row-major [feature][column] planes the same
// instruction touched 32 cache lines) -- and a 64-key tile of one head is four such blocks for the attention core's DMA.
__host__ __device__ inline size_t attn_v_off(int f, size_t col, int d) { return ((col >> 4) * (size_t)d + f) * 16 + (col & 15); }

struct AttnPlanes {
  _Float16 *qh, *ql, *kh, *kl;   // [nhead][T][32] head-major, Q pre-scaled by log2(e)/sqrt(d)
  _Float16 *vth, *vtl;           // [tp/16][nhead*32][16] blocked transposed (attn_v_off)
  const int* cu;                 // [nseg + 1]
  int nseg, t_total, tp;
  // device [4]: multipliers applied when the planes are written -- [0] Q: log2(e)/sqrt(d) * 2^-ek,
  // [1] K: 2^ek, [2] V: 2^ev -- and [3] = 2^-ev, applied to the attention output
  // (k_plane_scales, attention.hip)
  const float* scales;
};

// Shared by attention.hip and xenc.hip (the fused cross-encoder chains write the planes themselves):
// padded column count of the transposed V planes, carving of the planes out of a workspace of
// spr_attn_workspace_bytes(t, nseg, nhead, 32) bytes, zeroing of the gap / tail columns, and the
// attention core over planes that are already in place.  `mode`: 1 split-fp16, 2 single pass.
size_t attn_tp(int t, int nseg);
int attn_carve_planes(void* ws, size_t ws_bytes, int t, int nseg, int d, AttnPlanes& pl);
int attn_zero_gaps(const AttnPlanes& pl, int d, hipStream_t stream);
// o_tiles != nullptr: `out` is a TILED token tensor of the fused chains (xenc.hip; o_tiles[s] = chain tiles in front
// of segment s); only where attn_core_tiled_ok(mode).
int attn_core_on_planes(const AttnPlanes& pl, const int* kv_seg, int max_len_host, int nhead, float* out,
                        int o_stride, int mode, hipStream_t stream, const int* o_tiles = nullptr);
bool attn_core_tiled_ok(int mode);
int attn_mode();   // 1 split-fp16, 0 exact f32, 2 single-pass fp16

// Split-fp16 GEMM  planes <- x [m, k] . w [n, k]^T + bias  for the n output
// features [f0, f0 + n) of the packed in-projection (0..255 = Q, 256..511 = K,
// 512..767 = V; d_model = 256, head_dim = 32).  n and f0 are multiples of 256.
// a_parts / w_parts: absmax partials of x and w (launch_absmax).
int launch_inproj_planes(const float* x, int m, int k, const float* w, int n, const float* bias, int f0,
                         const AttnPlanes& planes, const float* a_parts, int n_aparts, const float* w_parts,
                         hipStream_t stream);

// Plain GEMM out = x w^T + bias with pre-measured operand ranges (nullptr in exact-f32 mode).
int launch_linear_ranged(const float* x, int m, int k, const float* w, int n, const float* bias, float* out,
                         const float* a_parts, const float* w_parts, hipStream_t stream);
int gemm_mode();   // 1 = split-fp16, 0 = exact f32

// One problem of a grouped NT GEMM (linear.hip, launch_gemm_grouped): element offsets into the
// shared A / B / C bases, sizes, and the running tile count (exclusive end) of the 1-D grid.
struct GemmGroup {
  long long a_off, b_off, c_off;
  int m, n, tile_end, pad;
};
void gemm_group_tile(int max_n, int* bm, int* bn);
// epilogue codes of the grouped GEMM (continue the SPR_ACT_* numbering of spr.h)
constexpr int kEpiScale = 3, kEpiAffinity = 4;
int launch_gemm_grouped(const float* a, int k, const float* b, float* c, const GemmGroup* groups_dev,
                        int total_tiles, int max_n, const float* a_parts, const float* w_parts, int epi_mode,
                        const float* epi, hipStream_t stream);

}  // namespace spr
