// Fused row chains of the cross-encoder (inference): everything of a pre-norm layer that is not
// the attention core runs in TWO kernels per layer, and no intermediate leaves the chip.
//
// Behaviour contract: TransformerCrossEncoderLayer.forward_pre + TransformerCrossEncoder.forward
//   /root/reference/src/models/transformer/transformers.py:184-245 (layer), :45-80 (stack, final norm)
// for the configuration every shipped experiment uses (pre_norm, sa/ca_val_has_pos_emb, sine
// positional embedding, dropout 0, ReLU, d_model 256 = 8 heads x 32).
//
// Every operator between two attention cores is ROW LOCAL (projection, bias, residual add, LayerNorm,
// positional-embedding add, ReLU), so a wave can own 32 tokens and push them through the whole chain:
//   chain A (behind the self attention)   o -> out_proj + x -> x' (stored) -> norm2 + pos -> in_proj -> planes
//   chain B (behind the cross attention)  o -> out_proj + x -> norm3 -> linear1 -> ReLU -> linear2 + x'
//                                           -> x'' (stored) -> norm1 of the NEXT layer + pos -> in_proj -> planes
//                                           (last layer: -> final norm -> out)
//   prologue                               x -> norm1 + pos -> in_proj -> planes
// "planes" = the split-fp16 Q / K / V^T operand planes the attention core consumes (attn_planes.h).
//
// Kernel shape (k_xenc_chain): 4 waves per workgroup, ONE wave per SIMD with the whole 512-entry
// register file; a wave owns 32 tokens, a workgroup 128, persistent workgroups walk the token tiles.
// All products are computed TRANSPOSED, C^T[feature][token] = W X^T with v_mfma_f32_32x32x16_f16:
// the token is the LANE (l & 31), so every row reduction (LayerNorm) is in-lane plus one exchange with
// lane ^ 32, and the 32 x 32 accumulator tile of one product is -- after scaling and splitting, with
// no lane movement and no LDS -- the B operand of the next product (registers 8s..8s+7 = k-step s;
// the k order inside a step is the fixed permutation kperm() below, applied to the weights when they are
// laid out).  Activations therefore live in registers from the first load to the last store:
//   operand planes 2 x 64 VGPRs (hi, lo; K = 256), accumulators 128 (out_proj / linear2 tile) or 16.
// Weights are the only stream: pre-split ONCE per weight version into fp16 hi / lo MFMA fragments in
// exactly the order the chain consumes them (spr_xenc_prepare: 32 KiB chunks = 32 fragments of
// 1 KiB), brought to LDS by global_load_lds_dwordx4 (no VGPRs, no VALU) into a ring of four chunks
// shared by the four waves, three chunks ahead, retired with counted s_waitcnt vmcnt + one raw
// s_barrier per chunk.  A fragment is one conflict-free ds_read_b128 per lane and feeds three MFMAs
// (al.bh + ah.bl + ah.bh, the split-fp16 product of spr_common.h).
//
// Token tensors between the kernels of one forward are TILED (round 5).  With token = lane, a row-major [T, 256] f32
// tensor is read 16 bytes per lane at a 1-KiB lane stride: 32 cache lines per instruction, which one CU's vector-memory
// path serves at ~14 bytes per cycle -- 9 400 cycles per 128-KiB tile tensor, six to eight of them per tile, all exposed
// (one wave per SIMD), a quarter of the chain's time.  So every tensor that only the chains and the attention core touch
// -- the residual stream x, the attention output o, a copy of the positional embedding -- is kept as
//   [chain tile][wave][piece k = 0..31][lane] x 16 bytes,   piece k of lane (token r, half h) = features 8 k + 4 h .. + 3
// i.e. exactly the registers of load_c() in the order the lanes hold them: every load and store is one fully coalesced
// 1-KiB wave instruction.  The prologue converts x and pos once; the attention core writes o in this form itself
// (k_attn_s, o_tiles); only the stack's input and its final output are row-major.  Rows past a cloud's end are computed
// and stored like any other (they start as copies of the cloud's last row and are never read back as tokens).
//
// Operand scales are STATIC: every tensor inside a chain has a data-independent bound
//   |LayerNorm(x)| <= sqrt(d - 1) max|gamma| + max|beta|,   |x W^T + b| <= bound(x) max_row ||W||_1 + max|b|,
//   |attention output| <= bound(V),   |pos| <= pos_bound
// evaluated on the host when the weights are prepared.  The bounds are loose by 2^4..2^8; the split
// keeps an absolute error of 2^-40 of the SCALED maximum, so the results stay at fp32 rounding level
// (spr_common.h, split_pk_s) and nothing can overflow.  No range is measured or handed over at run time.
#include <type_traits>
#include <vector>

#include "attn_planes.h"
#include "spr_common.h"

namespace spr {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int XD = 256;            // d_model
constexpr int XCHUNK = 32768;      // bytes per weight chunk (32 fragments of 1 KiB)
constexpr int XSLOTS = 4;          // ring depth (chunks)
constexpr int XAHEAD = 2;          // an acquire of chunk g (at step 8 of chunk g - 1) issues chunk g + XAHEAD
constexpr int XDMA = 8;            // LDS-DMA instructions per wave and chunk (32 KiB / 4 waves / 1 KiB)
constexpr int XTOK = 128;          // tokens per workgroup tile

// k order of a 16-deep k-step whose B operand is an accumulator tile (or is loaded in that shape):
// element j of lane half h <-> feature 16 S + 8 (j >> 2) + 4 h + (j & 3).
__host__ __device__ inline int kperm(int S, int h, int j) { return 16 * S + 8 * (j >> 2) + 4 * h + (j & 3); }

// One fused chain: device tables + host-evaluated constants.  Passed to the kernel by value.
struct ChainConsts {
  const unsigned char* w;               // weight stream: chunks of XCHUNK bytes in consumption order
  const float *bo, *b1, *b2, *bin;      // biases (the parameters themselves)
  const float *g_mid, *b_mid;           // LayerNorm in front of the feed-forward block (norm3)
  const float *g_tail, *b_tail;         // LayerNorm in front of the in-projection / final norm
  float eps_mid, eps_tail;
  float o_scale, res_o, un_o;           // out_proj: 2^ko, 2^(ko + kwo), 2^-(ko + kwo)
  float x2_scale, bs1, h_mul, res_f, un_f;   // FFN: 2^kx2, 2^(kx2 + kw1), 2^(kh - kx2 - kw1), 2^(kh + kw2), 2^-(kh + kw2)
  float xt_scale, bs_in, un_in;         // in_proj: 2^kxt, 2^(kxt + kwin), 2^-(kxt + kwin)
  float pmul[3];                        // plane multipliers (Q incl. log2(e)/sqrt(d), K, V)
  int nf;                               // d_ff / 32
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vm_any(int n) {   // n wave uniform, a multiple of 4 in 0..60
#define SPR_W(k) case k: wait_vm<k>(); break;
  switch (n) {
    SPR_W(4) SPR_W(8) SPR_W(12) SPR_W(16) SPR_W(20) SPR_W(24) SPR_W(28) SPR_W(32) SPR_W(36) SPR_W(40) SPR_W(44)
    SPR_W(48) SPR_W(52) SPR_W(56) SPR_W(60)
    default: wait_vm<0>(); break;
  }
#undef SPR_W
}

// LDS-DMA: 16 bytes per lane from base + off (per-lane byte offset) to LDS at dst_s + 16 * lane.
__device__ __forceinline__ void dma16(const void* base, unsigned off, unsigned dst_s) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(off), "s"(base), "s"(dst_s) : "memory", "m0");
}
// Four 1-KiB pieces behind ONE M0 / address setup: the instruction's immediate offset moves the global AND the LDS
// address (LDS address = M0 + offset + 16 lane), so piece q of a 4-KiB group is the same instruction with offset q KiB.
// dma_set() loads M0; dma_q<Q>() relies on nothing having written M0 since (the chunk loops hold no instruction that
// uses M0: LDS instructions of this architecture do not).
__device__ __forceinline__ void dma_set(unsigned dst_s) {
  asm volatile("s_mov_b32 m0, %0" : : "s"(dst_s) : "memory", "m0");
}
template <int Q>
__device__ __forceinline__ void dma_q(const void* base, unsigned off) {
  asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(off), "s"(base), "n"(Q * 1024) : "memory", "m0");
}
// (the trailing s_nop keeps the compiler's next instruction from overwriting the data registers of a
// wide store before it has read them)
__device__ __forceinline__ void store16(void* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store2(void* p, unsigned int v) {   // low 16 bits
  asm volatile("global_store_short %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store2hi(void* p, unsigned int v) {   // high 16 bits
  asm volatile("global_store_short_d16_hi %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// [32 tokens x 256 features] tile in the accumulator ("C") layout: v[b][e] = feature
// 32 b + (e & 3) + 8 (e >> 2) + 4 h of the lane's token.  The loads are VOLATILE: they are issued where
// they stand (a whole tile in one batch, long before its first use) and the compiler waits for them once,
// with one s_waitcnt -- which, blind to the LDS-DMAs queued behind them, also drains the weight ring: once
// per batch instead of once per table or row access (every other operand of the kernel comes from LDS).
__device__ __forceinline__ void load_c(const float* src, int tok, int h, f32x16 (&v)[8]) {
  const f32x4* p = reinterpret_cast<const f32x4*>(src + (size_t)tok * XD + 4 * h);
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 t = p[8 * b + 2 * g];
      v[b][4 * g + 0] = t[0];
      v[b][4 * g + 1] = t[1];
      v[b][4 * g + 2] = t[2];
      v[b][4 * g + 3] = t[3];
    }
}
__device__ __forceinline__ void store_c(float* __restrict__ dst, int tok, int h, bool valid, const f32x16 (&v)[8]) {
  if (!valid) return;
  float* p = dst + (size_t)tok * XD + 4 * h;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 t = {v[b][4 * g + 0], v[b][4 * g + 1], v[b][4 * g + 2], v[b][4 * g + 3]};
      *reinterpret_cast<f32x4*>(p + 32 * b + 8 * g) = t;
    }
}
// the same registers from / to a TILED tensor: wb4 = float4 index of the wave's 32-KiB block
constexpr size_t XWAVE4 = 2048;    // float4s per wave block (32 tokens x 256 features)
__device__ __forceinline__ void load_t(const float* src, size_t wb4, int lane, f32x16 (&v)[8]) {
  const f32x4* p = reinterpret_cast<const f32x4*>(src) + wb4 + lane;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 t = p[(4 * b + g) * 64];
      v[b][4 * g + 0] = t[0];
      v[b][4 * g + 1] = t[1];
      v[b][4 * g + 2] = t[2];
      v[b][4 * g + 3] = t[3];
    }
}
__device__ __forceinline__ void store_t(float* dst, size_t wb4, int lane, const f32x16 (&v)[8]) {
  f32x4* p = reinterpret_cast<f32x4*>(dst) + wb4 + lane;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 t = {v[b][4 * g + 0], v[b][4 * g + 1], v[b][4 * g + 2], v[b][4 * g + 3]};
      p[(4 * b + g) * 64] = t;
    }
}
// per-feature table (in LDS) in the same shape: t[g] = table[32 b + 8 g + 4 h .. + 3]
__device__ __forceinline__ void load_tab(const float* tab, int b, int h, f32x4 (&t)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) t[g] = *reinterpret_cast<const f32x4*>(tab + 32 * b + 8 * g + 4 * h);
}

// scaled split of one 32-feature block into the B fragments of its two k-steps
__device__ __forceinline__ void split_block(const f32x16& v, float scale, f16x8& h0, f16x8& l0, f16x8& h1,
                                            f16x8& l1) {
  unsigned int hu[8], lu[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) split_pk_s(v[2 * i], v[2 * i + 1], scale, hu[i], lu[i]);
  h0 = __builtin_bit_cast(f16x8, (u32x4){hu[0], hu[1], hu[2], hu[3]});
  l0 = __builtin_bit_cast(f16x8, (u32x4){lu[0], lu[1], lu[2], lu[3]});
  h1 = __builtin_bit_cast(f16x8, (u32x4){hu[4], hu[5], hu[6], hu[7]});
  l1 = __builtin_bit_cast(f16x8, (u32x4){lu[4], lu[5], lu[6], lu[7]});
}

// LayerNorm over the 256 features of the lane's token (two half rows: lanes l and l ^ 32).  Same
// operations as k_layernorm256 (two-pass mean / variance, (d rstd) gamma + beta); only the order of the
// two sums differs.  v is left untouched (it is also the residual).
__device__ __forceinline__ void ln_stats(const f32x16 (&v)[8], float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[b][e];
  s += __shfl_xor(s, 32, 64);
  mean = s / 256.0f;
  float ss = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float d = v[b][e] - mean;
      ss += d * d;
    }
  ss += __shfl_xor(ss, 32, 64);
  rstd = 1.0f / sqrtf(ss / 256.0f + eps);
}
__device__ __forceinline__ f32x16 ln_block(const f32x16& v, int b, int h, float mean, float rstd, const float* gamma,
                                           const float* beta) {
  f32x4 g[4], be[4];
  load_tab(gamma, b, h, g);
  load_tab(beta, b, h, be);
  f32x16 y;
#pragma unroll
  for (int e = 0; e < 16; ++e) y[e] = (v[e] - mean) * rstd * g[e >> 2][e & 3] + be[e >> 2][e & 3];
  return y;
}

// ---- chunk engine ----------------------------------------------------------------------------------
// One wave per SIMD: nothing else issues while this wave waits, and every instruction of the wave costs
// ~4 cycles of issue; what is not placed INSIDE the 32-cycle shadow of an MFMA is paid in full.  So the
// stream is laid out by hand and pinned with sched_barrier: a chunk is 16 steps; step i issues
//   MFMA (lo x hi) | side slice 2i | MFMA (hi x lo) | side slice 2i+1 | MFMA (hi x hi) | two fragment reads
//   for step i + XPF | (steps 8..15) one 1-KiB piece of the weight DMA
// where the "side" slices are the epilogue of the PREVIOUS chunk cut into 32 pieces (ReLU + split of the
// hidden chunk, plane stores, bias initialisation of the next accumulator, ...).  Fragment reads run three
// steps ahead through a ring of four register pairs carried from chunk to chunk: the last three steps of a
// chunk read the first fragments of the NEXT chunk, which is acquired (counted DMA wait + barrier) at step 8.
#define XSB() __builtin_amdgcn_sched_barrier(0)
#ifndef SPR_XENC_PF
#define SPR_XENC_PF 3            // fragment reads run this many steps ahead of their MFMAs
#endif
constexpr int XPF = SPR_XENC_PF;
constexpr int XRING = XPF < 4 ? 4 : 8;     // register pairs of the fragment ring (a power of two > XPF)
struct Carry {
  f16x8 h[XRING], l[XRING];
};
__device__ __forceinline__ void frag_ld(const unsigned char* slot, int lane, int i, f16x8& h, f16x8& l) {
  const f16x8* fr = reinterpret_cast<const f16x8*>(slot) + lane;
  h = fr[(2 * i) * 64];
  l = fr[(2 * i + 1) * 64];
}
__device__ __forceinline__ f32x16 mfma1(const f16x8& a, const f16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// LDS: [ring XSLOTS x XCHUNK] [tables] [cu]
constexpr int T_BO = 0, T_B2 = 256, T_GM = 512, T_BM = 768, T_GT = 1024, T_BT = 1280, T_BIN = 1536, T_B1 = 2304;
constexpr int T_PAD = 64;   // the bias tables are read one block past their end by the pipelined initialisations
__host__ __device__ inline size_t xenc_lds_bytes(int d_ff, int nseg) {
  return (size_t)XSLOTS * XCHUNK + (size_t)(T_B1 + d_ff + T_PAD + 2 * (nseg + 1)) * 4;
}

// Diagnostic build only (-DSPR_XENC_STAMP, scripts/xenc_timeline.py): shader-clock stamps of wave 0 of
// workgroup 0 at the stage boundaries of every tile, into a buffer nothing else reads.
#ifdef SPR_XENC_STAMP
__device__ unsigned long long g_xenc_stamps[64 * 64];
#define XSTAMP(k)                                                                                   \
  do {                                                                                              \
    if (blockIdx.x == 0 && tid == 0 && it < 64 && (stamp_on & 1))                                   \
      g_xenc_stamps[it * 64 + (k)] = __builtin_amdgcn_s_memtime();                                  \
  } while (0)
#define XSTAMP_REAL(k)                                                                              \
  do {                                                                                              \
    if (blockIdx.x == 0 && tid == 0 && it < 64 && (stamp_on & 1))                                   \
      g_xenc_stamps[it * 64 + (k)] = __builtin_amdgcn_s_memrealtime();                              \
  } while (0)
#define XSTAMP_IT(k)                                                                                \
  do {                                                                                              \
    if (blockIdx.x == 0 && tid == 0 && stamp_it < 64 && (stamp_on & 1))                             \
      g_xenc_stamps[stamp_it * 64 + (k)] = __builtin_amdgcn_s_memtime();                            \
  } while (0)
#else
#define XSTAMP(k) do { } while (0)
#define XSTAMP_REAL(k) do { } while (0)
#endif

// TAIL: 0 = x_out only, 1 = LayerNorm -> ln_out (final norm), 2 = LayerNorm + pos -> in-projection -> planes
// Layouts: HEAD chains read x_g TILED and o_g tiled (o_tiled) or row-major; the prologue (!HEAD) reads x_g and pos_g
// row-major and writes their tiled copies to xo_g and pos_t; xo_g is tiled unless TAIL == 0 (then it is the stack's
// row-major output); pos_t is read by every HEAD chain with an in-projection tail; ln_g is row-major.
template <bool HEAD, bool FFN, int TAIL>
__global__ __launch_bounds__(256, 1) void k_xenc_chain(ChainConsts c, const float* o_g, const float* x_g, float* xo_g,
                                                       const float* pos_g, float* pos_t, float* ln_g, AttnPlanes pl, int T,
                                                       int ntiles, int* tile_ctr, const int* __restrict__ tfirst_g,
                                                       int o_tiled, int stamp_on) {
  constexpr bool XO_TILED = TAIL != 0;
  extern __shared__ __align__(16) unsigned char ring[];
  float* tab = reinterpret_cast<float*>(ring + XSLOTS * XCHUNK);
  const int d_ff = 32 * c.nf;
  int* cu_s = reinterpret_cast<int*>(tab + T_B1 + d_ff + T_PAD);     // cu_seqlens [nseg + 1]
  int* tf_s = cu_s + pl.nseg + 1;                                      // tiles in front of every segment [nseg + 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const unsigned ring_s = (unsigned)(uintptr_t)ring;
  const int nch = (HEAD ? 8 : 0) + (FFN ? 2 * c.nf : 0) + (TAIL == 2 ? 24 : 0);   // chunks per tile
  // Tiles are dealt by a device counter (tile_ctr, zeroed by the host before the launch): workgroups that run
  // slower (or start later) take fewer tiles.  (A staggered start -- four phase groups per XCD, so that the
  // chip does not load its tile operands in one burst -- was measured and changed nothing: the operand phases
  // are bound by what ONE CU's vector-memory path moves, 128 KB per tile tensor at ~25 GB/s, not by HBM.)
  __shared__ int s_tile[2];
  // ---- parameter tables and cu_seqlens into LDS (the steady state reads its operands from LDS only) ----
  for (int i = tid; i < 256; i += 256) {
    tab[T_BO + i] = HEAD ? c.bo[i] : 0.f;
    tab[T_B2 + i] = FFN ? c.b2[i] : 0.f;
    tab[T_GM + i] = FFN ? c.g_mid[i] : 0.f;
    tab[T_BM + i] = FFN ? c.b_mid[i] : 0.f;
    tab[T_GT + i] = TAIL >= 1 ? c.g_tail[i] : 0.f;
    tab[T_BT + i] = TAIL >= 1 ? c.b_tail[i] : 0.f;
  }
  for (int i = tid; i < 768; i += 256) tab[T_BIN + i] = TAIL == 2 ? c.bin[i] : 0.f;
  for (int i = tid; i < d_ff + T_PAD; i += 256) tab[T_B1 + i] = (FFN && i < d_ff) ? c.b1[i] : 0.f;
  for (int i = tid; i <= pl.nseg; i += 256) {
    cu_s[i] = pl.cu[i];
    tf_s[i] = tfirst_g[i];
  }
  __syncthreads();
  ntiles = tf_s[pl.nseg];                // (the host only knows an upper bound)

  // ---- weight ring -----------------------------------------------------------------------------------
  // Chunk g lives in slot g % 4.  acquire() -- at step 8 of chunk g - 1 -- makes chunk g readable (every wave
  // waits for its own eight DMA pieces of it, then the barrier) and arms the pieces of chunk g + 2 for the
  // slot every wave has left (that of chunk g - 2); they are issued one per step in steps 8..15.  The stream
  // simply runs on past the last tile (positions wrap: valid memory, never read; drained before the exit),
  // so the counts are constants.
  int i_pos = 0, i_slot = 0;              // issue side: stream position, slot
  int a_slot = 0;
  int s_cnt = 0;                          // counted (asm) stores since the last acquire
  unsigned dma_dst = 0, dma_off = 0, dma_off4 = 0;      // armed pieces: LDS address, per-lane source offsets (+ 4 KiB)
  auto arm = [&]() {
    dma_dst = __builtin_amdgcn_readfirstlane(ring_s + (unsigned)i_slot * XCHUNK + (unsigned)wave * (XDMA * 1024));
    dma_off = (unsigned)i_pos * XCHUNK + (unsigned)wave * (XDMA * 1024) + (unsigned)lane * 16;
    dma_off4 = dma_off + 4096;
    if (++i_pos == nch) i_pos = 0;
    if (++i_slot == XSLOTS) i_slot = 0;
  };
  // piece j of the armed chunk (j is a constant after unrolling): pieces 0 and 4 set M0 for their group of four (the
  // wait state M0 needs in front of an LDS-DMA is the s_nop), the others are ONE instruction
  auto piece = [&](int j) __attribute__((always_inline)) {
#ifdef SPR_XENC_STAMP
    if (stamp_on & 2) return;             // ablation: no weight DMA (garbage results)
#endif
    switch (j) {
      case 0: dma_set(dma_dst); asm volatile("s_nop 0"); dma_q<0>(c.w, dma_off); break;
      case 1: dma_q<1>(c.w, dma_off); break;
      case 2: dma_q<2>(c.w, dma_off); break;
      case 3: dma_q<3>(c.w, dma_off); break;
      case 4: dma_set(dma_dst + 4096); asm volatile("s_nop 0"); dma_q<0>(c.w, dma_off4); break;
      case 5: dma_q<1>(c.w, dma_off4); break;
      case 6: dma_q<2>(c.w, dma_off4); break;
      default: dma_q<3>(c.w, dma_off4); break;
    }
  };
  auto acquire = [&]() __attribute__((always_inline)) -> const unsigned char* {
    // younger than the last piece of chunk g_acq: the eight pieces of chunk g_acq + 1 and the counted stores
    // since the previous acquire (rounded DOWN: an under-count only waits longer)
    if (s_cnt >= 32) wait_vm<40>();
    else if (s_cnt >= 4) wait_vm<12>();
    else wait_vm<8>();
#ifdef SPR_XENC_STAMP
    if (!(stamp_on & 4))                  // ablation: no barrier (garbage results)
#endif
    __builtin_amdgcn_s_barrier();
    arm();
    s_cnt = 0;
    const unsigned char* p = ring + a_slot * XCHUNK;
    if (++a_slot == XSLOTS) a_slot = 0;
    return p;
  };
  // first tile; staggered start
  if (tid == 0) s_tile[0] = atomicAdd(tile_ctr, 1);
  __syncthreads();
  int tile = s_tile[0];
  if (tile >= ntiles) return;
  // prologue: chunks 0, 1, 2 in flight, chunk 0 acquired, its first fragments read
#pragma unroll 1
  for (int q = 0; q < 3; ++q) {
    arm();
#pragma unroll
    for (int j = 0; j < XDMA; ++j) piece(j);
  }
  Carry cy;
  const unsigned char* slot = ring;
  wait_vm<16>();
  __builtin_amdgcn_s_barrier();
  a_slot = 1;
#pragma unroll
  for (int i = 0; i < XPF; ++i) frag_ld(slot, lane, i, cy.h[i], cy.l[i]);
  // chunk with accumulator `acc`, activation planes xh / xl (one fragment per step), side slices side(0..31).
  // SWAP = false: acc[feature][token] (weights are the A operand); true: acc[token][feature] (activations are the A
  // operand: the lane is a FEATURE and holds 16 tokens -- the V blocks, whose planes are stored transposed)
  int stamp_it = 0;
  int step_stamp = -1;       // diagnostic build: >= 0 = stamp every step of the next chunk at step_stamp + i
  auto chunk_f = [&](auto swap_tag, f32x16& acc, const f16x8 (&xh)[16], const f16x8 (&xl)[16], auto&& side)
      __attribute__((always_inline)) {
    constexpr bool SW = decltype(swap_tag)::value;
    const unsigned char* nxt = slot;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#ifdef SPR_XENC_STAMP
      if (step_stamp >= 0) XSTAMP_IT(step_stamp + i);
#endif
      if (i == 8) nxt = acquire();
      acc = SW ? mfma1(xh[i], cy.l[i & (XRING - 1)], acc) : mfma1(cy.l[i & (XRING - 1)], xh[i], acc);
      XSB();
      side(2 * i);
      XSB();
      acc = SW ? mfma1(xl[i], cy.h[i & (XRING - 1)], acc) : mfma1(cy.h[i & (XRING - 1)], xl[i], acc);
      XSB();
      side(2 * i + 1);
      XSB();
      acc = SW ? mfma1(xh[i], cy.h[i & (XRING - 1)], acc) : mfma1(cy.h[i & (XRING - 1)], xh[i], acc);
      XSB();
      if (i + XPF < 16) frag_ld(slot, lane, i + XPF, cy.h[(i + XPF) & (XRING - 1)], cy.l[(i + XPF) & (XRING - 1)]);
      else frag_ld(nxt, lane, i + XPF - 16, cy.h[(i + XPF) & (XRING - 1)], cy.l[(i + XPF) & (XRING - 1)]);
      if (i >= 8) piece(i - 8);
      XSB();
    }
    slot = nxt;
  };
  // linear2 chunk: y[i >> 1] += W2 fragment(i) . hidden k-step (i & 1)
  auto chunk_g = [&](f32x16 (&y)[8], const f16x8 (&hh)[2], const f16x8 (&hl)[2], auto&& side)
      __attribute__((always_inline)) {
    const unsigned char* nxt = slot;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#ifdef SPR_XENC_STAMP
      if (step_stamp >= 0) XSTAMP_IT(step_stamp + i);
#endif
      if (i == 8) nxt = acquire();
      y[i >> 1] = mfma1(cy.l[i & (XRING - 1)], hh[i & 1], y[i >> 1]);
      XSB();
      side(2 * i);
      XSB();
      y[i >> 1] = mfma1(cy.h[i & (XRING - 1)], hl[i & 1], y[i >> 1]);
      XSB();
      side(2 * i + 1);
      XSB();
      y[i >> 1] = mfma1(cy.h[i & (XRING - 1)], hh[i & 1], y[i >> 1]);
      XSB();
      if (i + XPF < 16) frag_ld(slot, lane, i + XPF, cy.h[(i + XPF) & (XRING - 1)], cy.l[(i + XPF) & (XRING - 1)]);
      else frag_ld(nxt, lane, i + XPF - 16, cy.h[(i + XPF) & (XRING - 1)], cy.l[(i + XPF) & (XRING - 1)]);
      if (i >= 8) piece(i - 8);
      XSB();
    }
    slot = nxt;
  };
  auto no_side = [](int) __attribute__((always_inline)) {};
  // accumulator <- bias table block (scaled), one group of four per call (g = 0..3)
  auto bias_group = [&](f32x16& a, const float* tb, float sc, int g) __attribute__((always_inline)) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(tb + 8 * g + 4 * h);
    a[4 * g + 0] = t[0] * sc;
    a[4 * g + 1] = t[1] * sc;
    a[4 * g + 2] = t[2] * sc;
    a[4 * g + 3] = t[3] * sc;
  };

#pragma unroll 1
  for (int it = 0; tile < ntiles; ++it) {
    stamp_it = it;
    if (tid == 0) s_tile[(it + 1) & 1] = atomicAdd(tile_ctr, 1);       // the next tile (read at the end of this one)
    // Tiles never straddle a segment (cloud): tile = (segment, 128-token slice of it).  A wave's 32 tokens then sit in
    // 32 consecutive, 32-byte aligned columns of the transposed V planes (one segment, vstart is a multiple of 16).
    int sg;
    {
      int lo = 0, hi = pl.nseg;                        // largest s with tf[s] <= tile (skips empty segments)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tf_s[mid] <= tile) lo = mid; else hi = mid;
      }
      sg = lo;
    }
    const int seg_beg = cu_s[sg], seg_end = cu_s[sg + 1];
    const int tok_w = seg_beg + (tile - tf_s[sg]) * XTOK + wave * 32;     // first token of the wave
    const int tok = tok_w + r;
    const bool valid = tok < seg_end;
    const bool wave_valid = tok_w < seg_end;               // lane 0 of the wave is valid: its stores do issue
    const int tokc = valid ? tok : seg_end - 1;
    const size_t wb4 = ((size_t)tile * 4 + wave) * XWAVE4;       // the wave's block of the tiled tensors
    (void)T;
    f32x16 v[8];
    f16x8 ph[16], pw[16];
    XSTAMP(0);
    XSTAMP_REAL(31);

    if constexpr (HEAD) {
      // ---- x' = o Wo^T + bo + x ------------------------------------------------------------------
      if (o_tiled) load_t(o_g, wb4, lane, v);
      else load_c(o_g, tokc, h, v);
#pragma unroll
      for (int b = 0; b < 8; ++b) split_block(v[b], c.o_scale, ph[2 * b], pw[2 * b], ph[2 * b + 1], pw[2 * b + 1]);
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_group(v[b], tab + T_BO + 32 * b, c.res_o, g);
      XSTAMP(1);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        chunk_f(std::false_type{}, v[b], ph, pw, no_side);
        if (b == 0) XSTAMP(2);
      }
      XSTAMP(3);
      {
        f32x16 t_x[8];
        load_t(x_g, wb4, lane, t_x);
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[b][e] = v[b][e] * c.un_o + t_x[b][e];
      }
      // the new residual stream (FFN: parked there, re-read behind the loop)
      if constexpr (XO_TILED) store_t(xo_g, wb4, lane, v);
      else store_c(xo_g, tok, h, valid, v);
      // The 32 stores are younger than every DMA piece in flight (the wait for x just drained the ring): the first
      // acquire of the feed-forward loop would otherwise wait for them as well.
      if constexpr (FFN) s_cnt += 32;
      XSTAMP(4);
    } else {
      load_c(x_g, tokc, h, v);
      store_t(xo_g, wb4, lane, v);          // prologue: the tiled copy of x the chains read
    }

    if constexpr (FFN) {
      // ---- x'' = x' + linear2(relu(linear1(norm3(x')))) -------------------------------------------
      // stream order F0, F1, G0, F2, G1, ..., F(nf-1), G(nf-2), G(nf-1): the ReLU + split of hidden chunk c
      // runs as side slices of F(c+1); two linear1 accumulators A / B alternate (nf is even)
      f32x16 y[8];
      {
        float mean, rstd;
        ln_stats(v, c.eps_mid, mean, rstd);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const f32x16 t = ln_block(v[b], b, h, mean, rstd, tab + T_GM, tab + T_BM);
          split_block(t, c.x2_scale, ph[2 * b], pw[2 * b], ph[2 * b + 1], pw[2 * b + 1]);
        }
      }
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias_group(y[b], tab + T_B2 + 32 * b, c.res_f, g);
      f32x16 A, B;
      unsigned int hu[8], lu[8];
      f16x8 hh[2], hl[2];
      const float* b1t = tab + T_B1;
#pragma unroll
      for (int g = 0; g < 4; ++g) bias_group(A, b1t, c.bs1, g);
      XSTAMP(5);
      // side of F(c): slices 0..7 ReLU + split of the previous accumulator P (hidden chunk c - 1), slices
      // 8..11 re-initialise P with the bias of hidden chunk c + 1 (the table is padded past d_ff)
      // (a slice must fit the ~24 issue cycles an MFMA leaves free: the split of one register pair is four half-rate
      // v_fma_mix*_f16 + the two maxima = 40 cycles, so it is cut in two -- hi plane in slice 2 q, lo plane in 2 q + 1)
      float ra = 0.f, rb = 0.f;
      auto relu_split = [&](f32x16& P, int cnext, int s) __attribute__((always_inline)) {
        if (s < 16) {
          const int q = s >> 1;
          if (!(s & 1)) {
            ra = fmaxf(P[2 * q], 0.f);
            rb = fmaxf(P[2 * q + 1], 0.f);
            split_pk_s_hi(ra, rb, c.h_mul, hu[q]);
          } else {
            split_pk_s_lo(ra, rb, c.h_mul, hu[q], lu[q]);
          }
        } else if (s < 20) {
          bias_group(P, b1t + 32 * cnext, c.bs1, s - 16);
        }
      };
      auto pack_h = [&]() __attribute__((always_inline)) {
        hh[0] = __builtin_bit_cast(f16x8, (u32x4){hu[0], hu[1], hu[2], hu[3]});
        hl[0] = __builtin_bit_cast(f16x8, (u32x4){lu[0], lu[1], lu[2], lu[3]});
        hh[1] = __builtin_bit_cast(f16x8, (u32x4){hu[4], hu[5], hu[6], hu[7]});
        hl[1] = __builtin_bit_cast(f16x8, (u32x4){lu[4], lu[5], lu[6], lu[7]});
      };
      // F0 (side: bias of hidden chunk 1 into B)
      chunk_f(std::false_type{}, A, ph, pw, [&](int s) __attribute__((always_inline)) {
        if (s < 4) bias_group(B, b1t + 32, c.bs1, s);
      });
      XSTAMP(6);
      const int npair = c.nf / 2 - 1;
#pragma unroll 1
      for (int j = 0; j < npair; ++j) {
        const int c1 = 2 * j + 1;
        chunk_f(std::false_type{}, B, ph, pw, [&](int s) __attribute__((always_inline)) { relu_split(A, c1 + 1, s); });   // F(c1), A = hidden c1-1
        pack_h();
        chunk_g(y, hh, hl, no_side);                                                                    // G(c1 - 1)
        chunk_f(std::false_type{}, A, ph, pw, [&](int s) __attribute__((always_inline)) { relu_split(B, c1 + 2, s); });   // F(c1 + 1)
        pack_h();
        chunk_g(y, hh, hl, no_side);                                                                    // G(c1)
        if (j == 0) XSTAMP(7);
      }
      {
        const int c1 = c.nf - 1;                                                                        // last pair
#ifdef SPR_XENC_STAMP
        step_stamp = 15;
#endif
        chunk_f(std::false_type{}, B, ph, pw, [&](int s) __attribute__((always_inline)) { relu_split(A, c1 + 1, s); });
        pack_h();
#ifdef SPR_XENC_STAMP
        step_stamp = 32;
#endif
        chunk_g(y, hh, hl, no_side);                                                                    // G(nf - 2)
#ifdef SPR_XENC_STAMP
        XSTAMP(48);
        step_stamp = -1;
#endif
#pragma unroll
        for (int s = 0; s < 8; ++s) split_pk_s(fmaxf(B[2 * s], 0.f), fmaxf(B[2 * s + 1], 0.f), c.h_mul, hu[s], lu[s]);
        pack_h();
        chunk_g(y, hh, hl, no_side);                                                                    // G(nf - 1)
      }
      XSTAMP(8);
      {
        f32x16 t_x[8];
        if constexpr (XO_TILED) load_t(xo_g, wb4, lane, t_x);             // x' again (this lane stored it)
        else load_c(xo_g, tokc, h, t_x);
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[b][e] = y[b][e] * c.un_f + t_x[b][e];
      }
      if constexpr (TAIL == 2) store_t(xo_g, wb4, lane, v);
      if constexpr (TAIL == 0) store_c(xo_g, tok, h, valid, v);
      XSTAMP(9);
    }

    if constexpr (TAIL == 1) {
      float mean, rstd;
      ln_stats(v, c.eps_tail, mean, rstd);
#pragma unroll
      for (int b = 0; b < 8; ++b) v[b] = ln_block(v[b], b, h, mean, rstd, tab + T_GT, tab + T_BT);
      store_c(ln_g, tok, h, valid, v);
    }
    if constexpr (TAIL == 2) {
      // ---- Q / K / V planes of norm(x) + pos ------------------------------------------------------
      {
        float mean, rstd;
        ln_stats(v, c.eps_tail, mean, rstd);
        f32x16 t_p[8];
        if constexpr (HEAD) {
          load_t(pos_t, wb4, lane, t_p);
        } else {
          load_c(pos_g, tokc, h, t_p);
          store_t(pos_t, wb4, lane, t_p);   // prologue: the tiled copy of the positional embedding
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          f32x16 t = ln_block(v[b], b, h, mean, rstd, tab + T_GT, tab + T_BT);
#pragma unroll
          for (int e = 0; e < 16; ++e) t[e] += t_p[b][e];
          split_block(t, c.xt_scale, ph[2 * b], pw[2 * b], ph[2 * b + 1], pw[2 * b + 1]);
        }
      }
      XSTAMP(10);
      const int vcol_w = attn_vstart_of(seg_beg, sg) + (tok_w - seg_beg);       // column of the wave's first token
      // 24 feature blocks (8 heads of Q, K, V); the epilogue of block fb - 1 (scale, split, plane stores) and
      // the bias of block fb + 1 run as side slices of block fb; accumulators A / B alternate.  The V blocks
      // (16..23) are computed transposed (chunk_f SWAP): their accumulator holds 16 tokens of one feature.
      const float* bint = tab + T_BIN;
      f32x16 A, B;
      unsigned int hu[8], lu[8];
#pragma unroll
      for (int g = 0; g < 4; ++g) bias_group(A, bint, c.bs_in, g);
      // bias of block fbn, a quarter per call: per register (Q / K) or per lane (V)
      auto bias_next = [&](f32x16& P, int fbn, int q) __attribute__((always_inline)) {
        if (fbn >= 16) {
          const float t = bint[32 * fbn + r] * c.bs_in;
          P[4 * q + 0] = t;
          P[4 * q + 1] = t;
          P[4 * q + 2] = t;
          P[4 * q + 3] = t;
        } else {
          bias_group(P, bint + 32 * fbn, c.bs_in, q);
        }
      };
      // epilogue slices of accumulator P = feature block fbp (kind WHICH: 0 Q, 1 K, 2 V), then bias of block fbn
      auto epi = [&](auto which_tag, f32x16& P, int fbp, int fbn, int s) __attribute__((always_inline)) {
        constexpr int WHICH = decltype(which_tag)::value;
        const int head = fbp & 7;
        if (s < 16) {
          // split of register pair q: hi plane in slice 2 q, lo plane in 2 q + 1 (a slice must fit the issue cycles an
          // MFMA leaves free).  un_in is a power of two: folded into the plane multiplier, same bits.
          const float pm = c.un_in * c.pmul[WHICH];
          const int q = s >> 1;
          if (!(s & 1)) split_pk_s_hi(P[2 * q], P[2 * q + 1], pm, hu[q]);
          else split_pk_s_lo(P[2 * q], P[2 * q + 1], pm, hu[q], lu[q]);
          // (round 5: the V planes keep every 16-token group in the order [0-3, 8-11, 4-7, 12-15] -- attn_planes.h,
          // attn_vperm -- which is exactly what a lane holds: (feature l & 31, half h) has tokens {0..3, 8..11} + 4 h of
          // the wave's first 16 and {16..19, 24..27} + 4 h of its second 16 as packed pairs.  The four
          // v_permlane32_swaps per plane that used to gather 8 consecutive tokens are gone.)
        } else if (s >= 22 && s < 26) {
          bias_next(P, fbn, s - 22);
        } else if (s >= 18 && s < 22) {            // stores behind the acquire of step 8 (slice 16): counted there
#ifdef SPR_XENC_STAMP
          if (stamp_on & 8) return;                // ablation: no plane stores (garbage results)
#endif
          const unsigned int* src = s < 20 ? hu : lu;
          const int o4 = 4 * (s & 1);
          if constexpr (WHICH < 2) {
            // head-major [head][token][32]: the lane's 16 features of the head as 32 contiguous bytes at
            // d' = 16 h + e.  The order of d inside a head is free as long as Q and K agree (both are
            // written here) -- the scores sum over it.
            if (valid) {
              _Float16* pb = WHICH == 0 ? (s < 20 ? pl.qh : pl.ql) : (s < 20 ? pl.kh : pl.kl);
              const size_t row = ((size_t)head * pl.t_total + tokc) * 32 + 16 * h + 8 * (s & 1);
              store16(pb + row, (u32x4){src[o4], src[o4 + 1], src[o4 + 2], src[o4 + 3]});
            }
          } else {
            // transposed planes [feature][token column]: piece (s & 1) = the lane's eight tokens {0..3, 8..11} + 4 h of
            // the 16-token group (s & 1) = one 16-byte chunk of the permuted row head * 32 + (l & 31); written when the
            // chunk's first token exists (the rest of a partly valid chunk are finite values of the clamped last token,
            // inside the gap in front of the next segment)
            const int t0 = tok_w + 16 * (s & 1) + 4 * h;
            if (t0 < seg_end) {
              _Float16* pb = s < 20 ? pl.vth : pl.vtl;
              // (blocked planes, attn_v_off: the wave's instruction writes the head's 1-KiB block of this 16-column group)
              const size_t o0 = attn_v_off(head * 32 + r, (size_t)(vcol_w + 16 * (s & 1) + 8 * h), XD);
              store16(pb + o0, (u32x4){src[o4], src[o4 + 1], src[o4 + 2], src[o4 + 3]});
            }
          }
        }
      };
      // stores a wave really issued since the last acquire.  Q / K blocks: four (all under the same lane mask).  V
      // blocks: the second piece of each plane exists only when the wave has more than 16 tokens (its store sits
      // behind its own execz branch) -- counting it for a ragged wave would let acquire() release a slot with up to
      // two DMA pieces of the acquired chunk still in flight (an over-count waits for LESS).
      auto count = [&](bool vblock) __attribute__((always_inline)) {
        if (wave_valid) s_cnt += (vblock && tok_w + 16 >= seg_end) ? 2 : 4;
      };
      // block 0 (side: bias of block 1 into B)
      chunk_f(std::false_type{}, A, ph, pw, [&](int s) __attribute__((always_inline)) {
        if (s < 4) bias_group(B, bint + 32, c.bs_in, s);
      });
      XSTAMP(11);
      // blocks fb (odd, into B) and fb + 1 (into A); which_tag = kind of the blocks whose epilogues run beside them
      auto pair = [&](auto which_tag, auto sw1, auto sw2, int fb) __attribute__((always_inline)) {
        constexpr bool VB = decltype(which_tag)::value == 2;
        chunk_f(sw1, B, ph, pw, [&](int s) __attribute__((always_inline)) { epi(which_tag, A, fb - 1, fb + 1, s); });
        count(VB);
        chunk_f(sw2, A, ph, pw, [&](int s) __attribute__((always_inline)) { epi(which_tag, B, fb, fb + 2, s); });
        count(VB);
      };
      const std::integral_constant<int, 0> kQ{};
      const std::integral_constant<int, 1> kK{};
      const std::integral_constant<int, 2> kV{};
      const std::false_type nsw{};
      const std::true_type sw{};
#pragma unroll 1
      for (int fb = 1; fb < 8; fb += 2) pair(kQ, nsw, nsw, fb);        // blocks 1..8, epilogues of blocks 0..7 (Q)
      XSTAMP(12);
#pragma unroll 1
      for (int fb = 9; fb < 14; fb += 2) pair(kK, nsw, nsw, fb);       // blocks 9..14, epilogues of blocks 8..13 (K)
      pair(kK, nsw, sw, 15);                                           // blocks 15 and 16 (the first V block)
      XSTAMP(13);
#pragma unroll 1
      for (int fb = 17; fb < 22; fb += 2) pair(kV, sw, sw, fb);        // blocks 17..22, epilogues of blocks 16..21 (V)
      chunk_f(sw, B, ph, pw, [&](int s) __attribute__((always_inline)) { epi(kV, A, 22, 23, s); });
      count(true);
#pragma unroll
      for (int s = 0; s < 32; ++s) epi(kV, B, 23, 23, s);              // last block: nothing to hide behind
      count(true);
      XSTAMP(14);
    }
    tile = s_tile[(it + 1) & 1];   // written before this tile's first barrier: ordered by the barriers since
  }
  wait_vm<0>();   // the chunks issued past the end of the stream must have landed before the LDS is released
}

// ---- weight preparation ------------------------------------------------------------------------------
// F kind: W [n, 256] (row stride ld): feature block fb -> chunk c0 + fb * cstride; fragment (S, plane)
// at 2 S + plane; lane l holds W[32 fb + (l & 31)][kperm(S, l >> 5, j)] * wscale, j = 0..7.
__global__ __launch_bounds__(256) void k_xenc_wprep_f(const float* __restrict__ W, int ld, int nblocks, float wscale,
                                                      unsigned char* __restrict__ out, int c0, int cstride) {
  const int gid = blockIdx.x * 256 + threadIdx.x;     // (fb, S, lane)
  if (gid >= nblocks * 16 * 64) return;
  const int l = gid & 63, S = (gid >> 6) & 15, fb = gid >> 10;
  const float* row = W + (size_t)(32 * fb + (l & 31)) * ld;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = row[kperm(S, l >> 5, j)];
  unsigned int hu[4], lu[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pk_s(v[2 * i], v[2 * i + 1], wscale, hu[i], lu[i]);
  unsigned char* chunk = out + (size_t)(c0 + fb * cstride) * XCHUNK;
  *reinterpret_cast<u32x4*>(chunk + (size_t)(2 * S) * 1024 + l * 16) = (u32x4){hu[0], hu[1], hu[2], hu[3]};
  *reinterpret_cast<u32x4*>(chunk + (size_t)(2 * S + 1) * 1024 + l * 16) = (u32x4){lu[0], lu[1], lu[2], lu[3]};
}
// G kind: W2 [256, ff]: hidden chunk ch -> chunk c0 + ch * cstride; fragment (fb, s, plane) at (fb 2 + s) 2 + plane;
// lane l holds W2[32 fb + (l & 31)][32 ch + kperm(s, l >> 5, j)] * wscale.
__global__ __launch_bounds__(256) void k_xenc_wprep_g(const float* __restrict__ W, int ld, int nchunks, float wscale,
                                                      unsigned char* __restrict__ out, int c0, int cstride) {
  const int gid = blockIdx.x * 256 + threadIdx.x;     // (ch, fb, s, lane)
  if (gid >= nchunks * 16 * 64) return;
  const int l = gid & 63, s = (gid >> 6) & 1, fb = (gid >> 7) & 7, ch = gid >> 10;
  const float* row = W + (size_t)(32 * fb + (l & 31)) * ld + 32 * ch;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = row[kperm(s, l >> 5, j)];
  unsigned int hu[4], lu[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pk_s(v[2 * i], v[2 * i + 1], wscale, hu[i], lu[i]);
  unsigned char* chunk = out + (size_t)(c0 + ch * cstride) * XCHUNK;
  const int fi = (fb * 2 + s) * 2;
  *reinterpret_cast<u32x4*>(chunk + (size_t)fi * 1024 + l * 16) = (u32x4){hu[0], hu[1], hu[2], hu[3]};
  *reinterpret_cast<u32x4*>(chunk + (size_t)(fi + 1) * 1024 + l * 16) = (u32x4){lu[0], lu[1], lu[2], lu[3]};
}

// max |w| and the largest row L1 norm of a [rows, cols] tensor: one workgroup per job.
struct StatJob {
  const float* w;
  int rows, cols;
};
__global__ __launch_bounds__(256) void k_xenc_stats(const StatJob* __restrict__ jobs, float* __restrict__ out) {
  __shared__ float sh[8];
  const StatJob j = jobs[blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float amax = 0.f, l1max = 0.f;
  for (int row = wave; row < j.rows; row += 4) {
    float s = 0.f;
    for (int cidx = lane; cidx < j.cols; cidx += 64) {
      const float a = fabsf(j.w[(size_t)row * j.cols + cidx]);
      s += a;
      amax = fmaxf(amax, a);
    }
    l1max = fmaxf(l1max, wave_sum(s));
  }
  amax = wave_max(amax);
  if (lane == 0) {
    sh[wave] = amax;
    sh[4 + wave] = l1max;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    out[2 * blockIdx.x + 1] = fmaxf(fmaxf(sh[4], sh[5]), fmaxf(sh[6], sh[7]));
  }
}

// ---- host plan -----------------------------------------------------------------------------------------
constexpr int kMaxLayers = 16;
constexpr uint32_t kPlanMagic = 0x58454e43u;   // "XENC"
struct Plan {
  uint32_t magic;
  int n_layers, d_ff, nhead;
  int has_final;
  ChainConsts prologue;              // norm1 of layer 0 + pos -> in_proj (self)
  ChainConsts a[kMaxLayers];         // behind the self attention
  ChainConsts b[kMaxLayers];         // behind the cross attention
  const float* scales_self[kMaxLayers];    // device [4] plane multipliers of the attention cores
  const float* scales_cross[kMaxLayers];
};

int chunks_prologue() { return 24; }
int chunks_a() { return 8 + 24; }
int chunks_b(int nf, bool tail_inproj) { return 8 + 2 * nf + (tail_inproj ? 24 : 0); }
size_t stream_chunks(int n_layers, int nf) {
  size_t n = chunks_prologue();
  for (int l = 0; l < n_layers; ++l) n += chunks_a() + chunks_b(nf, l + 1 < n_layers);
  return n;
}
constexpr size_t kStatsBytes = 64 * 1024;     // stat jobs + results + plane multipliers

// tiles in front of every segment: tf[s] = sum_{s' < s} ceil(len_s' / XTOK), tf[nseg] = number of tiles
__global__ void k_xenc_tiles(const int* __restrict__ cu, int nseg, int* __restrict__ tf) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int t = 0;
  for (int s = 0; s < nseg; ++s) {
    tf[s] = t;
    t += (cu[s + 1] - cu[s] + XTOK - 1) / XTOK;
  }
  tf[nseg] = t;
}

template <bool HEAD, bool FFN, int TAIL>
int launch_chain(const ChainConsts& c, const float* o, const float* x, float* xo, const float* pos, float* pos_t,
                 float* ln, const AttnPlanes& pl, int T, int* tile_ctr, const int* tfirst, int o_tiled,
                 hipStream_t stream) {
  auto kern = k_xenc_chain<HEAD, FFN, TAIL>;
  const size_t lds = xenc_lds_bytes(32 * c.nf, pl.nseg);
  SPR_REQUIRE(lds + 64 <= 160 * 1024, "xenc: d_ff %d with %d segments does not fit the LDS tables", 32 * c.nf, pl.nseg);
  if (int rc = ensure_dyn_lds((const void*)kern, 160 * 1024 - 64)) return rc;
  const int ntiles = cdiv(T, XTOK) + pl.nseg;       // upper bound (tiles do not straddle segments); exact count on the device
  int grid = device_cu_count();
  grid = grid < ntiles ? grid : ntiles;
  int stamp_on = 0;
#ifdef SPR_XENC_STAMP
  {   // SPR_XENC_STAMP_WHICH = "A", "B" or "P": only launches of that chain kind stamp (default: every launch)
    const char* e = getenv("SPR_XENC_STAMP_WHICH");
    const char kind = !HEAD ? 'P' : (FFN ? 'B' : 'A');
    const char* et = getenv("SPR_XENC_STAMP_TAIL");      // "0", "1" or "2": additionally only that tail
    stamp_on = (e == nullptr || e[0] == kind) && (et == nullptr || et[0] - '0' == TAIL) ? 1 : 0;
    const char* ea = getenv("SPR_XENC_ABL");             // ablation bits: 2 = no weight DMA, 4 = no barrier, 8 = no plane stores
    if (ea != nullptr) stamp_on |= atoi(ea) & 14;
  }
#endif
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, c, o, x, xo, pos, pos_t, ln, pl, T, ntiles, tile_ctr, tfirst,
                     o_tiled, stamp_on);
  SPR_LAUNCH_CHECK();
  return 0;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_xenc_prepared_bytes(int n_layers, int d_ff) {
  if (n_layers < 1 || n_layers > kMaxLayers || d_ff < 64 || d_ff % 64 != 0) return 0;
  return kStatsBytes + stream_chunks(n_layers, d_ff / 32) * (size_t)XCHUNK;
}

extern "C" size_t spr_xenc_plan_bytes(void) { return sizeof(Plan); }

// layer_ptrs_host: SPR_XENC_PTRS_PER_LAYER device pointers per layer, in the order
//   self_attn.in_proj_weight, .in_proj_bias, .out_proj.weight, .out_proj.bias,
//   multihead_attn.in_proj_weight, .in_proj_bias, .out_proj.weight, .out_proj.bias,
//   linear1.weight, .bias, linear2.weight, .bias, norm1.weight, .bias, norm2.weight, .bias, norm3.weight, .bias
extern "C" int spr_xenc_prepare(const void* const* layer_ptrs_host, const float* eps_host, int n_layers, int d_model,
                                int nhead, int d_ff, const float* final_g, const float* final_b, float final_eps,
                                float pos_bound, void* prepared, size_t prepared_bytes, void* plan_host,
                                size_t plan_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(d_model == XD && nhead == 8, "xenc: d_model must be 256 with 8 heads (got %d, %d)", d_model, nhead);
  SPR_REQUIRE(n_layers >= 1 && n_layers <= kMaxLayers && d_ff >= 64 && d_ff % 64 == 0,
              "xenc: bad n_layers / d_ff (%d, %d): d_ff must be a multiple of 64", n_layers, d_ff);
  SPR_REQUIRE(layer_ptrs_host && eps_host && prepared && plan_host, "xenc_prepare: null argument");
  SPR_REQUIRE(prepared_bytes >= spr_xenc_prepared_bytes(n_layers, d_ff) && plan_bytes >= sizeof(Plan),
              "xenc_prepare: buffers too small");
  SPR_REQUIRE((final_g == nullptr) == (final_b == nullptr), "xenc_prepare: final norm needs weight and bias");
  SPR_REQUIRE(pos_bound >= 0.f, "xenc_prepare: pos_bound must be >= 0");
  const int nf = d_ff / 32;
  enum { SA_W, SA_B, SA_OW, SA_OB, CA_W, CA_B, CA_OW, CA_OB, W1, B1, W2, B2, N1G, N1B, N2G, N2B, N3G, N3B, NPTR };
  static_assert(NPTR == SPR_XENC_PTRS_PER_LAYER, "pointer table layout");
  auto P = [&](int l, int which) { return (const float*)layer_ptrs_host[l * NPTR + which]; };
  for (int l = 0; l < n_layers; ++l)
    for (int w = 0; w < NPTR; ++w) SPR_REQUIRE(P(l, w) != nullptr, "xenc_prepare: layer %d parameter %d is null", l, w);

  // ---- statistics: max |.| and max row L1 of every tensor (in-projections per Q / K / V block) ------
  // per layer: 0-2 sa_w blocks, 3-5 sa_b blocks, 6 sa_ow, 7 ca_w x3 .. 9, 10-12 ca_b blocks, 13 ca_ow,
  //            14 w1, 15 b1, 16 w2, 17 n1g, 18 n1b, 19 n2g, 20 n2b, 21 n3g, 22 n3b ; then final g, b
  constexpr int JPL = 23;
  std::vector<StatJob> jobs;
  for (int l = 0; l < n_layers; ++l) {
    for (int q = 0; q < 3; ++q) jobs.push_back({P(l, SA_W) + (size_t)q * XD * XD, XD, XD});
    for (int q = 0; q < 3; ++q) jobs.push_back({P(l, SA_B) + (size_t)q * XD, 1, XD});
    jobs.push_back({P(l, SA_OW), XD, XD});
    for (int q = 0; q < 3; ++q) jobs.push_back({P(l, CA_W) + (size_t)q * XD * XD, XD, XD});
    for (int q = 0; q < 3; ++q) jobs.push_back({P(l, CA_B) + (size_t)q * XD, 1, XD});
    jobs.push_back({P(l, CA_OW), XD, XD});
    jobs.push_back({P(l, W1), d_ff, XD});
    jobs.push_back({P(l, B1), 1, d_ff});
    jobs.push_back({P(l, W2), XD, d_ff});
    jobs.push_back({P(l, N1G), 1, XD});
    jobs.push_back({P(l, N1B), 1, XD});
    jobs.push_back({P(l, N2G), 1, XD});
    jobs.push_back({P(l, N2B), 1, XD});
    jobs.push_back({P(l, N3G), 1, XD});
    jobs.push_back({P(l, N3B), 1, XD});
  }
  if (final_g) {
    jobs.push_back({final_g, 1, XD});
    jobs.push_back({final_b, 1, XD});
  }
  const size_t njobs = jobs.size();
  // head of the prepared buffer: jobs | results | plane multipliers
  unsigned char* base = (unsigned char*)prepared;
  StatJob* jobs_dev = (StatJob*)base;
  float* res_dev = (float*)(base + 16 * 1024);
  float* scales_dev = (float*)(base + 32 * 1024);            // [2 n_layers][4]
  SPR_REQUIRE(njobs * sizeof(StatJob) <= 16 * 1024 && njobs * 2 * sizeof(float) <= 16 * 1024, "xenc_prepare: too many tensors");
  SPR_HIP_CHECK(hipMemcpyAsync(jobs_dev, jobs.data(), njobs * sizeof(StatJob), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(k_xenc_stats, dim3((unsigned)njobs), dim3(256), 0, stream, jobs_dev, res_dev);
  SPR_LAUNCH_CHECK();
  std::vector<float> res(njobs * 2);
  SPR_HIP_CHECK(hipMemcpyAsync(res.data(), res_dev, njobs * 2 * sizeof(float), hipMemcpyDeviceToHost, stream));
  SPR_HIP_CHECK(hipStreamSynchronize(stream));
  auto amax = [&](int l, int j) { return res[2 * (l * JPL + j)]; };
  auto l1 = [&](int l, int j) { return res[2 * (l * JPL + j) + 1]; };
  for (float f : res) SPR_REQUIRE(f == f && f < 3.0e38f, "xenc_prepare: non-finite parameter");

  // 2^k on the host; the single exponents are clamped to +-60 (pow2_exp_for), their sums must stay normal floats
  bool exp_ok = true;
  auto pow2f = [&](int k) {
    if (k < -120 || k > 120) exp_ok = false;
    return ldexpf(1.0f, k);
  };
  Plan* plan = (Plan*)plan_host;
  memset(plan, 0, sizeof(Plan));
  plan->n_layers = n_layers;
  plan->d_ff = d_ff;
  plan->nhead = nhead;
  plan->has_final = final_g != nullptr;
  unsigned char* wbase = base + kStatsBytes;
  size_t chunk_at = 0;
  std::vector<float> scales_host((size_t)2 * n_layers * 4);
  const float lnk = sqrtf((float)(XD - 1)) * 1.0001f;
  auto ln_bound = [&](float gmax, float bmax) { return lnk * gmax + bmax; };

  auto prep_f = [&](const float* W, int ld, int nblocks, int kw, int c0, int cstride) -> int {
    hipLaunchKernelGGL(k_xenc_wprep_f, dim3(cdiv((long)nblocks * 1024, 256)), dim3(256), 0, stream, W, ld, nblocks,
                       pow2f(kw), wbase, c0, cstride);
    SPR_LAUNCH_CHECK();
    return 0;
  };
  auto prep_g = [&](const float* W, int ld, int nchunks, int kw, int c0, int cstride) -> int {
    hipLaunchKernelGGL(k_xenc_wprep_g, dim3(cdiv((long)nchunks * 1024, 256)), dim3(256), 0, stream, W, ld, nchunks,
                       pow2f(kw), wbase, c0, cstride);
    SPR_LAUNCH_CHECK();
    return 0;
  };
  // in-projection tail: input bound xb (norm + pos); jw = first stat job of the weight blocks, jb of the bias blocks
  auto setup_inproj = [&](ChainConsts& c, int l, int jw, int jb, const float* W, const float* B, const float* g,
                          const float* be, float eps, float gmax, float bmax, float* sc, float* o_bound,
                          int c0) -> int {
    const float xb = ln_bound(gmax, bmax) + pos_bound;
    const int kx = pow2_exp_for(xb);
    const float wmax = fmaxf(fmaxf(amax(l, jw), amax(l, jw + 1)), amax(l, jw + 2));
    const int kw = pow2_exp_for(wmax);
    c.g_tail = g;
    c.b_tail = be;
    c.eps_tail = eps;
    c.bin = B;
    c.xt_scale = pow2f(kx);
    c.bs_in = pow2f(kx + kw);
    c.un_in = pow2f(-kx - kw);
    // plane multipliers (k_plane_scales, attention.hip) from the bounds of the three blocks
    const float qb = xb * l1(l, jw) + amax(l, jb), kb = xb * l1(l, jw + 1) + amax(l, jb + 1),
                vb = xb * l1(l, jw + 2) + amax(l, jb + 2);
    const float qscale = 1.4426950408889634f / sqrtf(32.0f);
    const float qs = qb * qscale;
    int ek = 0;
    if (qs > 0.f && kb > 0.f && qs < 3.0e38f && kb < 3.0e38f) {
      int eq, ekk;
      frexpf(qs, &eq);
      frexpf(kb, &ekk);
      ek = ((eq - 1) - (ekk - 1)) >> 1;
      ek = ek < -60 ? -60 : (ek > 60 ? 60 : ek);
    }
    const int ev = pow2_exp_for(vb);
    sc[0] = qscale * pow2f(-ek);
    sc[1] = pow2f(ek);
    sc[2] = pow2f(ev);
    sc[3] = pow2f(-ev);
    c.pmul[0] = sc[0];
    c.pmul[1] = sc[1];
    c.pmul[2] = sc[2];
    *o_bound = pow2f(15 - ev);
    return prep_f(W, XD, 24, kw, c0, 1);
  };
  auto setup_head = [&](ChainConsts& c, int l, int jw, const float* W, const float* B, float o_bound, int c0) -> int {
    const int ko = pow2_exp_for(o_bound), kw = pow2_exp_for(amax(l, jw));
    c.bo = B;
    c.o_scale = pow2f(ko);
    c.res_o = pow2f(ko + kw);
    c.un_o = pow2f(-ko - kw);
    return prep_f(W, XD, 8, kw, c0, 1);
  };

  float o_bound = 0.f;
  // prologue: norm1 of layer 0 + pos -> self in-projection
  {
    ChainConsts& c = plan->prologue;
    c.w = wbase + chunk_at * XCHUNK;
    c.nf = nf;
    if (int rc = setup_inproj(c, 0, 0, 3, P(0, SA_W), P(0, SA_B), P(0, N1G), P(0, N1B), eps_host[0], amax(0, 17),
                              amax(0, 18), &scales_host[0], &o_bound, (int)chunk_at))
      return rc;
    chunk_at += chunks_prologue();
  }
  for (int l = 0; l < n_layers; ++l) {
    // chain A: self out_proj + residual, norm2 + pos, cross in-projection
    {
      ChainConsts& c = plan->a[l];
      c.w = wbase + chunk_at * XCHUNK;
      c.nf = nf;
      if (int rc = setup_head(c, l, 6, P(l, SA_OW), P(l, SA_OB), o_bound, (int)chunk_at)) return rc;
      if (int rc = setup_inproj(c, l, 7, 10, P(l, CA_W), P(l, CA_B), P(l, N2G), P(l, N2B), eps_host[3 * l + 1],
                                amax(l, 19), amax(l, 20), &scales_host[(size_t)(2 * l + 1) * 4], &o_bound,
                                (int)chunk_at + 8))
        return rc;
      chunk_at += chunks_a();
    }
    // chain B: cross out_proj + residual, norm3, feed forward + residual, then norm1 of layer l + 1 + pos and
    // its self in-projection (or the final norm)
    {
      ChainConsts& c = plan->b[l];
      c.w = wbase + chunk_at * XCHUNK;
      c.nf = nf;
      if (int rc = setup_head(c, l, 13, P(l, CA_OW), P(l, CA_OB), o_bound, (int)chunk_at)) return rc;
      const float x2b = ln_bound(amax(l, 21), amax(l, 22));
      const int kx2 = pow2_exp_for(x2b), kw1 = pow2_exp_for(amax(l, 14));
      const float hb = x2b * l1(l, 14) + amax(l, 15);
      const int kh = pow2_exp_for(hb), kw2 = pow2_exp_for(amax(l, 16));
      c.g_mid = P(l, N3G);
      c.b_mid = P(l, N3B);
      c.eps_mid = eps_host[3 * l + 2];
      c.b1 = P(l, B1);
      c.b2 = P(l, B2);
      c.x2_scale = pow2f(kx2);
      c.bs1 = pow2f(kx2 + kw1);
      c.h_mul = pow2f(kh - kx2 - kw1);
      c.res_f = pow2f(kh + kw2);
      c.un_f = pow2f(-kh - kw2);
      // stream order F0, F1, G0, F2, G1, ..., F(nf-1), G(nf-2), G(nf-1)  (k_xenc_chain, feed-forward block)
      const int fbase = (int)chunk_at + 8;
      if (int rc = prep_f(P(l, W1), XD, 1, kw1, fbase, 1)) return rc;                                    // F0
      if (int rc = prep_f(P(l, W1) + (size_t)32 * XD, XD, nf - 1, kw1, fbase + 1, 2)) return rc;          // F(c) at 2c - 1
      if (nf > 1)
        if (int rc = prep_g(P(l, W2), d_ff, nf - 1, kw2, fbase + 2, 2)) return rc;                        // G(c) at 2c + 2
      if (int rc = prep_g(P(l, W2) + (size_t)32 * (nf - 1), d_ff, 1, kw2, fbase + 2 * nf - 1, 1)) return rc;   // G(nf-1)
      if (l + 1 < n_layers) {
        if (int rc = setup_inproj(c, l + 1, 0, 3, P(l + 1, SA_W), P(l + 1, SA_B), P(l + 1, N1G), P(l + 1, N1B),
                                  eps_host[3 * (l + 1)], amax(l + 1, 17), amax(l + 1, 18),
                                  &scales_host[(size_t)(2 * (l + 1)) * 4], &o_bound, (int)chunk_at + 8 + 2 * nf))
          return rc;
      } else if (final_g) {
        c.g_tail = final_g;
        c.b_tail = final_b;
        c.eps_tail = final_eps;
      }
      chunk_at += chunks_b(nf, l + 1 < n_layers);
    }
  }
  SPR_REQUIRE(chunk_at == stream_chunks(n_layers, nf), "xenc_prepare: internal chunk count mismatch");
  SPR_REQUIRE(exp_ok, "xenc_prepare: parameter magnitudes out of the range the split arithmetic can scale");
  SPR_HIP_CHECK(hipMemcpyAsync(scales_dev, scales_host.data(), scales_host.size() * sizeof(float), hipMemcpyHostToDevice,
                               stream));
  SPR_HIP_CHECK(hipStreamSynchronize(stream));   // scales_host goes out of scope
  for (int l = 0; l < n_layers; ++l) {
    plan->scales_self[l] = scales_dev + (size_t)(2 * l) * 4;
    plan->scales_cross[l] = scales_dev + (size_t)(2 * l + 1) * 4;
  }
  plan->magic = kPlanMagic;
  return 0;
}

#ifdef SPR_XENC_STAMP
// diagnostic build: zero / read the stamp buffer (not part of include/spr.h)
extern "C" int spr_xenc_debug_stamps(unsigned long long* out_host, int clear) {
  if (clear) {
    static unsigned long long zeros[64 * 64];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_xenc_stamps), zeros, sizeof(zeros)) != hipSuccess;
  }
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_xenc_stamps), sizeof(unsigned long long) * 64 * 64) != hipSuccess;
}
#endif

// bytes of one tiled token tensor: an upper bound of the tile count (tiles do not straddle clouds) x 128 KiB; a
// row-major [t, 256] tensor (the attention output when the core cannot write tiles) fits as well
static size_t xenc_tiled_bytes(int t, int nseg) { return (size_t)(cdiv(t, XTOK) + nseg) * XTOK * XD * sizeof(float); }

extern "C" size_t spr_xenc_workspace_bytes(int t, int nseg) {
  if (t < 1 || nseg < 1) return 0;
  return spr_attn_workspace_bytes(t, nseg, 8, 32) + 4 * xenc_tiled_bytes(t, nseg) + 1024 +
         align_up((size_t)(nseg + 1) * sizeof(int), 256);
}

// x [t, 256] tokens of all clouds (packed), pos [t, 256] positional embedding, cu [nseg + 1],
// kv_self / kv_cross [nseg] key segment of every query segment; out [t, 256].
extern "C" int spr_xenc_forward(const void* plan_host, const float* x, const float* pos, const int* cu,
                                const int* kv_self, const int* kv_cross, int t, int nseg, int max_len_host,
                                float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const Plan* plan = (const Plan*)plan_host;
  SPR_REQUIRE(plan != nullptr && plan->magic == kPlanMagic, "xenc_forward: plan is not prepared");
  SPR_REQUIRE(x && pos && cu && kv_self && kv_cross && out, "xenc_forward: null argument");
  SPR_REQUIRE(t >= 1 && nseg >= 1 && max_len_host >= 1, "xenc_forward: bad sizes");
  SPR_REQUIRE((size_t)t * XD * 4 < (1ull << 32), "xenc_forward: too many tokens");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_xenc_workspace_bytes(t, nseg), "xenc_forward: workspace too small");
  const int mode = attn_mode();
  SPR_REQUIRE(gemm_mode() == 1 && (mode >= 1 && mode <= 4),
              "xenc_forward: needs the split-fp16 arithmetic (gemm mode 1, attention mode 1 or 2)");
  const size_t planes_bytes = spr_attn_workspace_bytes(t, nseg, 8, 32);
  AttnPlanes pl{};
  if (int rc = attn_carve_planes(ws, planes_bytes, t, nseg, XD, pl)) return rc;
  pl.cu = cu;
  pl.nseg = nseg;
  pl.t_total = t;
  pl.tp = (int)attn_tp(t, nseg);
  const size_t act = xenc_tiled_bytes(t, nseg);
  SPR_REQUIRE(act < (1ull << 32), "xenc_forward: too many tokens");
  float* obuf = (float*)((char*)ws + planes_bytes);            // attention output (tiled when the core can write tiles)
  float* xa = (float*)((char*)ws + planes_bytes + act);        // residual stream, tiled, ping
  float* xb = (float*)((char*)ws + planes_bytes + 2 * act);    //                          pong
  float* pos_t = (float*)((char*)ws + planes_bytes + 3 * act); // positional embedding, tiled
  int* ctr = (int*)((char*)ws + planes_bytes + 4 * act);       // one tile counter per chain launch
  SPR_HIP_CHECK(hipMemsetAsync(ctr, 0, 1024, stream));
  int* tfirst = (int*)((char*)ws + planes_bytes + 4 * act + 1024);
  hipLaunchKernelGGL(k_xenc_tiles, dim3(1), dim3(64), 0, stream, cu, nseg, tfirst);
  if (int rc = attn_zero_gaps(pl, XD, stream)) return rc;
  const int L = plan->n_layers;
  const int o_tiled = attn_core_tiled_ok(mode) ? 1 : 0;
  const int* o_tiles = o_tiled ? tfirst : nullptr;
  // prologue: planes of layer 0's self attention + the tiled copies of x (into xb) and pos
  if (int rc = launch_chain<false, false, 2>(plan->prologue, nullptr, x, xb, pos, pos_t, nullptr, pl, t, ctr++, tfirst, 0, stream)) return rc;
  const float* cur = xb;
  float* nxt = xa;
  for (int l = 0; l < L; ++l) {
    pl.scales = plan->scales_self[l];
    if (int rc = attn_core_on_planes(pl, kv_self, max_len_host, 8, obuf, XD, mode, stream, o_tiles)) return rc;
    if (int rc = launch_chain<true, false, 2>(plan->a[l], obuf, cur, nxt, nullptr, pos_t, nullptr, pl, t, ctr++, tfirst, o_tiled, stream)) return rc;
    cur = nxt;
    nxt = (nxt == xa) ? xb : xa;
    pl.scales = plan->scales_cross[l];
    if (int rc = attn_core_on_planes(pl, kv_cross, max_len_host, 8, obuf, XD, mode, stream, o_tiles)) return rc;
    if (l + 1 < L) {
      if (int rc = launch_chain<true, true, 2>(plan->b[l], obuf, cur, nxt, nullptr, pos_t, nullptr, pl, t, ctr++, tfirst, o_tiled, stream)) return rc;
      cur = nxt;
      nxt = (nxt == xa) ? xb : xa;
    } else if (plan->has_final) {
      if (int rc = launch_chain<true, true, 1>(plan->b[l], obuf, cur, nxt, nullptr, pos_t, out, pl, t, ctr++, tfirst, o_tiled, stream)) return rc;
    } else {
      if (int rc = launch_chain<true, true, 0>(plan->b[l], obuf, cur, out, nullptr, pos_t, nullptr, pl, t, ctr++, tfirst, o_tiled, stream)) return rc;
    }
  }
  return 0;
}
