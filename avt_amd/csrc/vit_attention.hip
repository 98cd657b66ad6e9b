// ViT spatial multi-head self-attention (timm Attention: softmax(q k^T * hd^-0.5) v, no mask, no dropout),
// forward and backward, head_dim 64, sequence S <= 208 (ViT-*/16 @224: S = 197), bf16 I/O, fp32 softmax.
//
// One workgroup per (frame, head); wave w owns the 16-row strip w of the sequence (13 waves for S = 197).
// The whole K / V (and Q / dO in backward) of one head fit in LDS, so the forward needs no online softmax and
// the backward recomputes P from the saved log-sum-exp.
// All products run on v_mfma_f32_16x16x32_bf16.  Scores are produced TRANSPOSED (S^T = K Q^T, keys along
// accumulator registers, queries along lanes) so that P^T / dS^T are already in the B-operand register layout
// of the following P V / dS K products; the second operand of those products (V^T, K^T, Q^T, dO^T) is gathered from
// the row-major LDS tiles with gfx950's transposing ds_read_b64_tr_b16, so nothing is ever transposed in memory.
//   qkv layout: [frames*S, 3*D] with columns [q | k | v], each head-major (timm reshape(N,S,3,H,hd)).
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
// Settled by A/B runs (records in profiles/; the switches themselves live on in tools/lab/avt_lab_hooks.diff):
//  * L2 policy: nt on the K / V / Q / dO tile loads costs 15 % (forward 708 -> 820 us at 2560 frames: the twelve heads of a frame read adjacent
//    128-byte pieces of the same qkv rows) and nt on the output stores more (forward 727 -> 945 us): default policy everywhere (profiles/r04_cache_policy.txt).
//  * The outputs leave in 16-byte pieces (round 5, see the forward kernel's last lines; profiles/r05d_attention_stores.txt).
//  * A start stagger of the persistent workgroups gains nothing (1845-1881 vs 1834-1845 us: the CUs are not in a harmful lockstep).
//  * The dK / dV of an item leave in chunk 0 of the NEXT item (packed, 16 registers that are free there: the new accumulators are not live before the
//    chunk's first dV product) instead of in the item's tail, next to the strip requests (-2.6 % per launch); barrier S2 inside chunk 0 and the tail in
//    front of the last chunk's dQ products measured no better (profiles/r05n_attention_boundary.txt).
#define AVT_ATTN_STG(p, v) (*(p) = (v))
#include "../../include/avt_hip.h"

namespace {

constexpr int HD = 64;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}



// row-major [R][64] bf16 tile, 128-B rows, 16-B chunk c of row r stored at c ^ swz8(r).  Two access patterns read these tiles:
//   * ds_read_b128 row fragments (16 rows x one chunk): conflict-free when the 8 row pairs of a 16-row group get 8 different
//     chunk positions (even / odd rows already sit in different 128-B halves of the 256-B bank row): any bijection of (r >> 1) & 7;
//   * ds_read_b64_tr_b16 transposing reads (8 rows x 32 B per half wave): the 4 row pairs of 8 consecutive rows must land on 4
//     different 32-B slots.  With the plain c ^ ((r >> 1) & 7) they shared two (rows r and r + 2 on the same banks: 25 % of the
//     backward's LDS cycles were bank-conflict cycles, profiles/r03a_pmc_sq.txt); p -> ((p & 3) << 1) | (p >> 2) satisfies both.
// (late round 3: the first "both patterns" permutation, p -> ((p & 3) << 1) | (p >> 2), turned out 2-way conflicted for the ds_read_b128 fragments --
// the hardware's lane grouping is not the one assumed above.  tools/lab/lds_swizzle_search.hip times all 8! bijections on the GPU: 3456 are clean for
// the b128 pattern, 1536 of those also for the transposing reads; this is the first of them, {0,2,4,6,5,7,1,3}.)
__device__ __forceinline__ int swz8(int r) { const int p = (r >> 1) & 7, hi = p >> 2; return ((((p & 3) ^ (hi << 1)) << 1) | hi); }
__device__ __forceinline__ int rm_off(int r, int c) { return r * 128 + ((c ^ swz8(r)) << 4); }

// Stage S rows x 64 columns (global row stride ld) into a swizzled row-major tile of RP rows (zero padded) and/or
// a transposed tile [64][TS] (element (d, r) at d*TS + r), zero padded to TP rows.
__device__ __forceinline__ void stage_head(const bf16_t* __restrict__ src, int ld, int S, char* rm, int RP, bf16_t* tr, int TS,
                                           int TP, int tid, int nthr) {
  int RMAX = RP > TP ? RP : TP;
  for (int idx = tid; idx < RMAX * 8; idx += nthr) {
    int r = idx >> 3, c = idx & 7;
    u32x4_t w = {0u, 0u, 0u, 0u};
    if (r < S) w = *(const u32x4_t*)(src + (size_t)r * ld + c * 8);
    if (rm && r < RP) *(u32x4_t*)(rm + rm_off(r, c)) = w;
    if (tr && r < TP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tr[(c * 8 + 2 * e) * TS + r] = (bf16_t)(w[e] & 0xffffu);
        tr[(c * 8 + 2 * e + 1) * TS + r] = (bf16_t)(w[e] >> 16);
      }
    }
  }
}


// Asynchronous variant for the row-major tiles: LDS-DMA (buffer_load ... lds, 16 B per lane, 8 rows x 128 B per wave
// instruction).  The LDS image is lane-linear, so the bank swizzle is applied to the SOURCE chunk index; rows >= S get an
// out-of-range offset and arrive as zeros.  All requests of a workgroup are in flight together; the caller waits once.
__device__ __forceinline__ void stage_head_dma(const bf16_t* src, int ld, int S, char* rm, int RP, int wave, int nwaves, int lane) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7FFFFFF0u, 0x00020000);
  for (int j = wave; j < RP / 8; j += nwaves) {
    const int r = j * 8 + (lane >> 3);
    const int c = (lane & 7) ^ swz8(r);
    uint32_t off = (uint32_t)(((size_t)r * (size_t)ld + (size_t)c * 8) * 2);
    if (r >= S) off = 0xFFFFFFF0u;
    char* dst = rm + __builtin_amdgcn_readfirstlane(j) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
  }
}

// B-operand fragments of a 16-row strip taken straight from global: lane (j = l&15, g = l>>4) holds
// X[row0 + j][ks*32 + g*8 .. +7], ks = 0, 1.  Rows >= S read as zero.
__device__ __forceinline__ void load_strip(const bf16_t* __restrict__ src, int ld, int S, int row0, int lane, bf16x8_t out[2]) {
  int r = row0 + (lane & 15), g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    union { u32x4_t u; bf16x8_t v; } t;
    t.u = (u32x4_t){0u, 0u, 0u, 0u};
    if (r < S) t.u = *(const u32x4_t*)(src + (size_t)r * ld + ks * 32 + g * 8);
    out[ks] = t.v;
  }
}
// A-operand fragment from a swizzled row-major LDS tile: lane (i = l&15, g) holds X[tile*16 + i][ks*32 + g*8 .. +7]
__device__ __forceinline__ bf16x8_t frag_rm(const char* rm, int tile, int ks, int lane) {
  int r = tile * 16 + (lane & 15), c = ks * 4 + (lane >> 4);
  return *(const bf16x8_t*)(rm + rm_off(r, c));
}
// A-operand fragment from a transposed LDS tile for the pair of 16-index tiles (2t, 2t+1):
// lane (i = l&15, g) holds Xt[dt*16 + i][32t + 4g .. +3] ++ Xt[dt*16 + i][32t + 16 + 4g .. +3]
__device__ __forceinline__ bf16x8_t frag_tr(const bf16_t* tr, int TS, int dt, int t, int lane) {
  const bf16_t* p = tr + (dt * 16 + (lane & 15)) * TS + 32 * t + 4 * (lane >> 4);
  union { bf16x8_t v; u32x2_t h[2]; } u;
  u.h[0] = *(const u32x2_t*)(p);
  u.h[1] = *(const u32x2_t*)(p + 16);
  return u.v;
}
// Same operand as frag_tr but gathered from a swizzled ROW-MAJOR tile with gfx950's transposing LDS read: within a
// 16-lane group, lane i supplies the address of block row (i>>2), columns 4*(i&3).., and receives column i of the
// [4 rows][16 cols] block -- i.e. X[32t + 4g + j][dt*16 + i], j = 0..3 (and the same 16 rows further for the upper half).
__device__ __forceinline__ bf16x8_t frag_tr_rm(const char* rm, int dt, int t, int lane) {
  const int g = lane >> 4, i16 = lane & 15;
  const int col = dt * 16 + (i16 & 3) * 4;
  union { bf16x8_t v; s16x4_t h[2]; } u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 32 * t + 16 * h + 4 * g + (i16 >> 2);
    const char* p = rm + rm_off(r, col >> 3) + (col & 7) * 2;
    u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  }
  return u.v;
}
// The same gather through inline assembly, for the K / V / Q / dO tiles that are filled by LDS-DMA: hipcc cannot tell that a
// __builtin_amdgcn_ds_read_tr16_b64 does not alias the buffer_load ... lds transfers in flight and puts s_waitcnt vmcnt(0) in
// front of it -- which here meant waiting, in the middle of an item, for the NEXT item's prefetched tiles (found in round 3).
// The result may only be used after lgkm_wait4() on it.
// Per-lane byte offsets of the lane's element for the four 16-column blocks dt (row 4g + (i>>2) of a 32-row group): the
// swizzle depends on the lane only ((row >> 1) & 7 = 2g + (i >> 3) for every t and h), so the tile, t and h are immediates.
__device__ __forceinline__ void tr_lane_offsets(int lane, uint32_t (&off)[4]) {
  const int g = lane >> 4, i16 = lane & 15;
  const int r = 4 * g + (i16 >> 2);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) off[dt] = (uint32_t)(rm_off(r, dt * 2 + ((i16 & 3) >> 1)) + (i16 & 1) * 8);
}
// fragment (dt, t) of the tile at byte offset TILE from the address the offsets were added to: rows 32t + 16h + ...
template <int TILE, int T>
__device__ __forceinline__ bf16x8_t frag_tr_na(uint32_t addr) {
  const u32x2_t lo = ds_read_tr_na<TILE + T * 4096>(addr), hi = ds_read_tr_na<TILE + T * 4096 + 2048>(addr);
  return tr_join(lo, hi);
}
// the four dt fragments of key / query pair t (t is a loop index of an unrolled loop: dispatched to the immediate forms)
template <int TILE, int NP>
__device__ __forceinline__ void frag4_tr_na(bf16x8_t (&f)[4], const uint32_t (&addr)[4], int t) {
  constexpr int T1 = 1 < NP ? 1 : 0, T2 = 2 < NP ? 2 : 0, T3 = 3 < NP ? 3 : 0, T4 = 4 < NP ? 4 : 0, T5 = 5 < NP ? 5 : 0, T6 = 6 < NP ? 6 : 0;   // t < NP always
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    switch (t) {
      case 0: f[dt] = frag_tr_na<TILE, 0>(addr[dt]); break;   case 1: f[dt] = frag_tr_na<TILE, T1>(addr[dt]); break;
      case 2: f[dt] = frag_tr_na<TILE, T2>(addr[dt]); break;  case 3: f[dt] = frag_tr_na<TILE, T3>(addr[dt]); break;
      case 4: f[dt] = frag_tr_na<TILE, T4>(addr[dt]); break;  case 5: f[dt] = frag_tr_na<TILE, T5>(addr[dt]); break;
      default: f[dt] = frag_tr_na<TILE, T6>(addr[dt]); break;
    }
  }
}
// wait for every LDS read of this wave; the fragments pass through the statement so that no consumer can be scheduled above it
__device__ __forceinline__ void lgkm_wait4(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ bf16x8_t pack_pair(f32x4_t a, f32x4_t b) {
  union { bf16x8_t v; uint32_t w[4]; } u;
  u.w[0] = pack2bf(a[0], a[1]); u.w[1] = pack2bf(a[2], a[3]);
  u.w[2] = pack2bf(b[0], b[1]); u.w[3] = pack2bf(b[2], b[3]);
  return u.v;
}
// sum / max over the 4 lane groups g (lanes sharing l&15)
__device__ __forceinline__ float gsum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float gmax(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }
// sum over the 16 lanes of a group (l&15) = one DPP row: four rotate-and-add steps on the VALU (v_add_f32 with a row_ror
// modifier) instead of four ds_bpermute round trips through the LDS crossbar -- the backward takes 48 such sums per item
__device__ __forceinline__ float lsum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}

// four such sums at once, written out: hipcc folds the rotate into the add (v_add_f32_dpp) only sometimes, else it emits a DPP move, a
// zero and an add per step.  The four independent chains also fill the two wait states a DPP read needs after the write of its source.
__device__ __forceinline__ void lsum16x4(f32x4_t& v) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  v = (f32x4_t){a, b, c, d};
}

// A wave's own strips of the NEXT item are requested by hand-written loads: the compiler does not track them, so no s_waitcnt of its own appears
// between the requests and the counted wait at the top of the next item.  With compiler-tracked loads (until late round 5) hipcc put
//   * backward: `s_waitcnt vmcnt(0)` right behind the lse load -- its `* log2(e)` was the first use of a loaded value -- so every wave sat out the full
//     memory latency in the tail of every item, before its dK / dV stores (tools/lab/attn_timeline.py: 9-13 k of an item's 44 k cycles); with the
//     multiplication moved away the wait reappeared in front of the first use at the loop top, now also covering the four stores just issued;
//   * forward: `s_waitcnt vmcnt(0)` on the loop's back edge (the Q strip is copied there), i.e. behind the output stores of every item, and one more in
//     front of the barrier (__syncthreads() = release fence).
// Rows past the sequence get an out-of-range offset and arrive as zeros (STRIP_OOB: beyond num_records, and + 64 of the instruction offset
// does not wrap around 32 bits as it would with the 0xFFFFFFF0 of the LDS-DMA requests).
constexpr uint32_t STRIP_OOB = 0x80000000u;
__device__ __forceinline__ u32x4_t raw_rsrc(const void* p) {
  const uint64_t a = (uint64_t)p;
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = 0x7FFFFFF0u; r[3] = 0x00020000u;
  return r;
}
template <int IMM>
__device__ __forceinline__ bf16x8_t strip_ld_na(u32x4_t rsrc, uint32_t voff, uint32_t soff) {
  bf16x8_t v;
  // s_nop 4: the hazard recogniser does not look into inline assembly, and a scalar operand may have just been written by a vector instruction
  // (v_readlane of a spilled SGPR, v_readfirstlane): 5 wait states before a memory instruction reads it.  Measured the hard way: without them the
  // V strip's scalar offset was occasionally stale and test_big_tile_gemm_and_attention_bit_reproducible failed.
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff), "n"(IMM));
  return v;
}
__device__ __forceinline__ float dword_ld_na(const float* base, uint32_t voff) {
  float v;
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(base));
  return v;
}

// Forward: persistent workgroups walking over (frame, head) items with the K / V tiles double-buffered in LDS: the tiles of
// item n+1 are requested (LDS-DMA) when item n starts computing, and this wave's Q strip of item n+1 right after, so the
// only exposed global latency is the very first item's.
template <int NKT, bool ALL_LIVE>
__global__ __launch_bounds__(64 * NKT) void vit_attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                float* __restrict__ lse, int S, int H, int items, float scale) {
  constexpr int NP = (NKT + 1) / 2;          // key-tile pairs
  constexpr int KP = NP * 32;                // padded key count
  constexpr int RM = KP * 128;               // one [KP][64] swizzled row-major tile
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][K | V] (V is consumed through transposing reads)
  const int D = H * HD, ld = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int q0 = wave * 16, g = lane >> 4;
  int item = blockIdx.x;
  // ONE register set for the Q strip: the next item's strip is requested into it as soon as this item's score products have read it, and nothing but
  // the counted wait at the loop top touches it in between (a second set would be copied into the first by the compiler -- in front of the wait)
  bf16x8_t bq[2];
  auto fetch_q = [&](const bf16_t* hb) __attribute__((always_inline)) {      // this wave's Q strip (B-operand layout) of the head slice at hb
    int lane_f = lane;                         // (opaque: the per-lane offset is recomputed here, not kept across the item)
    asm volatile("" : "+v"(lane_f));
    const int qf = q0 + (lane_f & 15);
    const uint32_t o = qf < S ? (uint32_t)((qf * ld + (lane_f >> 4) * 8) * 2) : STRIP_OOB;
    const u32x4_t r = raw_rsrc(hb);
    bq[0] = strip_ld_na<0>(r, o, 0u); bq[1] = strip_ld_na<64>(r, o, 0u);
  };
  uint32_t troff[4];
  tr_lane_offsets(lane, troff);
  if (item < items) {
    const bf16_t* base = qkv + (size_t)(item / H) * S * ld + (item % H) * HD;
    stage_head_dma(base + D, ld, S, smem, KP, wv, NKT, lane);
    stage_head_dma(base + 2 * D, ld, S, smem + RM, KP, wv, NKT, lane);
    fetch_q(base);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  int buf = 0;
  for (; item < items; item += gridDim.x, buf ^= 1) {
    const int frame = item / H, head = item % H;
    const char* Kb = smem + buf * 2 * RM;
    const char* Vb = Kb + RM;
    // K, V of this item were requested one item ago; younger than them are only the 2 Q-strip loads and the 3 (5) stores of the
    // previous item (nothing at the first item)
    // (ALL_LIVE: every wave's strip has at least one row inside the sequence, so every guarded load / store is issued; else
    // a wave may have skipped them and the count would be wrong -> wait for everything)
    // (wide stores: 2 + 3)
    // (late round 5: the Q strip, requested right behind the tiles, is waited for here too -- all but the previous item's 3 / 5 stores -- and the
    // barrier is a bare s_barrier: __syncthreads() brought a full `s_waitcnt vmcnt(0)` with it, i.e. the completion of those stores)
#define AVT_Q_LANDED(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bq[0]), "+v"(bq[1]) :: "memory")
    if (ALL_LIVE) AVT_Q_LANDED(3);
    else AVT_Q_LANDED(0);
#undef AVT_Q_LANDED
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // also: everyone is done with the other buffer (item n-1)
    const int nitem = item + gridDim.x < items ? item + gridDim.x : item;     // the last item re-requests itself (keeps the counts)
    const bf16_t* nbase = qkv + (size_t)(nitem / H) * S * ld + (nitem % H) * HD;
    {
      char* nb = smem + (buf ^ 1) * 2 * RM;
      stage_head_dma(nbase + D, ld, S, nb, KP, wv, NKT, lane);
      stage_head_dma(nbase + 2 * D, ld, S, nb + RM, KP, wv, NKT, lane);
    }
    f32x4_t st[2 * NP];
  #pragma unroll
    for (int kt = 0; kt < 2 * NP; ++kt) st[kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float mx = -3.0e38f;
  #pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x4_t a = {0.f, 0.f, 0.f, 0.f};
      a = mfma16(frag_rm(Kb, kt, 0, lane), bq[0], a);
      a = mfma16(frag_rm(Kb, kt, 1, lane), bq[1], a);
  #pragma unroll
      for (int r = 0; r < 4; ++r) {
        int key = kt * 16 + 4 * g + r;
        if (!ALL_LIVE || kt == NKT - 1) a[r] = (key < S) ? a[r] : -3.0e38f;     // ALL_LIVE: only the last key tile can hold keys >= S
        mx = fmaxf(mx, a[r]);
      }
      st[kt] = a;
    }
    // (the score products above were the last readers of the Q strip; the 11 wait states between an MFMA's operand read and a memory write
    // to the same registers are far exceeded by the request's latency)
    fetch_q(nbase);
    mx = gmax(mx);
    const float sl = scale * LOG2E;
    const float mxs = mx * sl;
    float sum = 0.f;
    // probabilities, packed to bf16 pair by pair as they are produced (the fp32 score registers die here: room for a second set of V
    // fragments below)
    bf16x8_t pb[NP];
  #pragma unroll
    for (int t = 0; t < NP; ++t) {
  #pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kt = 2 * t + u;
        if (kt < NKT) {
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            int key = kt * 16 + 4 * g + r;
            float pv = __builtin_amdgcn_exp2f(fmaf(st[kt][r], sl, -mxs));
            if (!ALL_LIVE || kt == NKT - 1) pv = (key < S) ? pv : 0.f;
            st[kt][r] = pv;
            sum += pv;
          }
        }
      }
      pb[t] = pack_pair(st[2 * t], st[2 * t + 1]);
    }
    sum = gsum(sum);
    f32x4_t o[4];
  #pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    uint32_t vaddr[4];                         // this item's V tile + the lane's offsets (4 adds per item)
    {
      const uint32_t vb = lds_addr32(Vb);
  #pragma unroll
      for (int dt = 0; dt < 4; ++dt) vaddr[dt] = vb + troff[dt];
    }
    // P V with the V fragments of key pair t + 1 requested before the products of pair t (two register sets)
    bf16x8_t vf[2][4];
    frag4_tr_na<0, NP>(vf[0], vaddr, 0);
  #pragma unroll
    for (int t = 0; t < NP; ++t) {
      if (t + 1 < NP) {
        frag4_tr_na<0, NP>(vf[(t + 1) & 1], vaddr, t + 1);
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(vf[t & 1][0]), "+v"(vf[t & 1][1]), "+v"(vf[t & 1][2]), "+v"(vf[t & 1][3]));   // the 8 reads just queued may still be out
      } else {
        lgkm_wait4(vf[t & 1][0], vf[t & 1][1], vf[t & 1][2], vf[t & 1][3]);
      }
  #pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma16(vf[t & 1][dt], pb[t], o[dt]);
    }
    const int q = q0 + (lane & 15);
    const float inv = 1.0f / sum;
    {
      // 16-byte stores: v_permlane16_swap exchanges the odd 16-lane rows of block dt with the even rows of block dt + 1, after which lane (i, g)
      // holds 8 consecutive columns of block dt + (g & 1): 64 contiguous bytes per output row and instruction instead of 32 (round 5: backward
      // 2142 -> 1983 us, forward 785 -> 775 us per launch at 2560 frames) ...
      u32x4_t st2[2];
#pragma unroll
      for (int dp = 0; dp < 4; dp += 2) {
        u32x2_t a, b;
        a[0] = pack2bf(o[dp][0] * inv, o[dp][1] * inv); a[1] = pack2bf(o[dp][2] * inv, o[dp][3] * inv);
        b[0] = pack2bf(o[dp + 1][0] * inv, o[dp + 1][1] * inv); b[1] = pack2bf(o[dp + 1][2] * inv, o[dp + 1][3] * inv);
        const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
        st2[dp >> 1] = (u32x4_t){r0[0], r1[0], r0[1], r1[1]};
      }
      // (whole 128-byte rows per instruction -- a further DPP row_ror:8 exchange between the two pieces -- measured no better: backward
      // 1996 -> 1965 us but three spilled registers, forward 790 -> 795 us, the step 910.9 vs 910.6 clips/s; profiles/r05d_attention_stores.txt)
      if (q < S) {
        bf16_t* orow = out + ((size_t)frame * S + q) * D + head * HD + (g >> 1) * 8;
        *(u32x4_t*)(orow + (g & 1) * 16) = st2[0];
        *(u32x4_t*)(orow + (2 + (g & 1)) * 16) = st2[1];
      }
    }
    if (q < S && g == 0) lse[((size_t)frame * H + head) * S + q] = mx * scale + __logf(sum);
  }
}

// ---- single-pass backward (round 4) -------------------------------------------------------------------------------------------
// The two-phase kernel of rounds 2-3 (one pass per query strip for dQ, one per key strip for dK / dV; tools/lab/avt_lab_hooks.diff) evaluated the scores and dP = dO V^T twice (once per query strip for dQ, once per key strip for
// dK / dV): 2 x 676 of its 2444 MFMAs per item and both softmax recomputations.  Here every (query tile, key tile) pair is evaluated
// ONCE, by the wave that owns the key strip (wave w = keys 16w .. 16w+15, K / V strips held in registers straight from global):
//     chunk c = queries 32c .. 32c+31:   S^T, dP^T (4 MFMAs per query tile)  ->  P, dS  ->  dV += P^T dO, dK += dS^T Q (8 MFMAs)
//                                        dS (bf16) -> LDS buffer c & 1, [key][32 queries], 72-byte rows
//     barrier c
//     dQ of the chunk's two query tiles = dS K, read back through transposing LDS reads; the 2 x 2 (query tile, half of the head dim)
//     products go to four different waves (h = 4c .. 4c+3 -> wave h mod NKT: every wave gets two of the 26 over an item), 14 MFMAs
//     each, while everybody else is already in chunk c+1.
// 1768 MFMAs per item instead of 2444, one softmax pass, 8 barriers per item instead of 2.  LDS: Q, dO, K tiles (84 KB; V never goes
// to LDS) + two dS buffers (32 KB) + per-query scalars and the bias staging = 131 KB.  Tiles are single-buffered: the rows of
// Q / dO that chunk c consumed are re-filled with the NEXT item's rows (LDS-DMA) right after barrier c, the K tile right after the
// item's first barrier -- it is needed at barrier 0 at the earliest.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// rows [8j, 8j + 8) of a head's [S][64] slice -> the swizzled row-major tile (one LDS-DMA instruction)
__device__ __forceinline__ void dma_rows8(__amdgpu_buffer_rsrc_t rsrc, int ld, int S, char* rm, int j, int lane) {
  const int r = j * 8 + (lane >> 3);
  const int c = (lane & 7) ^ swz8(r);
  uint32_t off = (uint32_t)(((size_t)r * (size_t)ld + (size_t)c * 8) * 2);
  if (r >= S) off = 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(rm + j * 1024), 16, off, 0, 0, 0);
}

// OT ("O tile", late round 5; chosen by the launcher when the LDS has room: H <= 21 at NKT = 13): the head's O rows get a row-major tile of their own
// (NKT * 16 rows, filled by LDS-DMA chunk by chunk like the Q / dO rows), and D[q] = sum_d dO[q,d] O[q,d] of a wave's strip is formed from the two LDS
// tiles after the item's first barrier instead of from 16 registers of global strips requested in the previous item's tail: those requests could not be
// issued early enough (no registers) and their latency sat exposed in front of the first barrier of every item (tools/lab/attn_timeline.py); price: a
// second barrier (S2) in front of chunk 0.
template <int NKT, bool ALL_LIVE, bool SCALED, bool OT>
__global__ __launch_bounds__(64 * NKT) void vit_attn_bwd1_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                                 const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                 bf16_t* __restrict__ dqkv, float* __restrict__ dbias,
                                                                 float* __restrict__ part, int S, int H, int items, float scale,
                                                                 const float* __restrict__ row_scale) {
  // row_scale != NULL (avt_vit_attn_bwd_scaled): row r of dqkv leaves multiplied by row_scale[2 r] -- the rstd of the LayerNorm folded into the qkv
  // projection (dY' = rstd o dY, include/avt_hip.h); the bias column sums stay those of the unscaled gradient
  constexpr int NP = (NKT + 1) / 2;          // 32-query chunks = key / query tile pairs
  constexpr int KP = NP * 32;                // rows of every LDS tile (zero padded)
  constexpr int RM = KP * 128;               // bytes of a row-major [KP][64] bf16 tile
  constexpr int DSP = 72;                    // dS buffer: [key][32 queries] bf16, 72-byte rows (4 consecutive rows x 32 B and 16 rows x 8 B hit distinct banks)
  constexpr int DSB = KP * DSP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* dOs = smem + RM;
  char* Ks = smem + 2 * RM;
  char* dSb = smem + 3 * RM;
  float* lse_s = (float*)(smem + 3 * RM + 2 * DSB);    // lse * log2(e)
  float* dq_s = lse_s + KP;                  // D[q] * scale,  D[q] = sum_d dO[q,d] O[q,d]
  float* bias_s = dq_s + KP;                 // [H][2*64] column sums of dq | dv, kept for the whole kernel (those of dk are zero: every row of dS sums to zero)
  float* stq_s = bias_s + H * 128;           // [NKT][64] this item's dq sums per query tile
  float* stv_s = stq_s + NKT * 64;           // [NP][64]  this item's dO sums per query pair (= the dv sums: every row of P sums to one)
  float* rs_s = stv_s + NP * 64;             // [KP] row_scale of this item's rows (read by the dQ products: query rows)
  char* Os = (char*)(rs_s + KP);             // OT: [NKT * 16][64] O rows, swizzled row-major like the Q / dO tiles
  constexpr int OTB = OT ? NKT * 16 * 128 : 0;
  int prev_head = -1;
  const int D = H * HD, ld = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = 64 * NKT;
  const int g = lane >> 4, i16 = lane & 15;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const float sl = scale * LOG2E;
  if (dbias) for (int i = tid; i < H * 128; i += nthr) bias_s[i] = 0.f;
  // key rows past the last wave's strip exist in the dS buffers (the dQ products walk whole 32-key pairs) but are never written: zero
  // them once -- they meet all-zero K rows, and 0 x (whatever bits the LDS held) must not be a NaN
  constexpr int PADW = (KP - NKT * 16) * (DSP / 4);          // 32-bit words of padding rows per buffer (none when the strips fill the pairs)
  if constexpr (PADW > 0) {
    for (int i = tid; i < 2 * PADW; i += nthr) {
      const int b = i / PADW, r = i % PADW;
      ((uint32_t*)(dSb + b * DSB + NKT * 16 * DSP))[r] = 0u;
    }
  }

  uint32_t tr0[4], trK[4];                   // per-lane addresses of the transposing reads: Q / dO tiles, K tile
  tr_lane_offsets(lane, tr0);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { tr0[dt] += lds_addr32(smem); trK[dt] = tr0[dt] + 2 * RM; }
  char* const dswr = dSb + (16 * wave + i16) * DSP + 8 * g;                                       // this lane's dS writes: key row, queries 4g..4g+3 of a tile
  const uint32_t dsrd = lds_addr32(dSb) + (uint32_t)((4 * g + (i16 >> 2)) * DSP + (i16 & 3) * 8);  // this lane's address in a transposing read of the buffer
  const int k0 = wave * 16, key = k0 + i16;  // own key strip; the same rows as a QUERY tile for the per-query scalars

  // own strips (B-operand layout, straight from global): K, V of this wave's keys; dO, O of the same rows as queries.  ONE register set each: the next
  // item's K / V strips are requested into bk / bv as soon as the last chunk's score products have read them, dO / O / the per-row scalars after the
  // last chunk's barrier; nothing but the counted wait at the loop top touches them in between (a second set would be copied into the first by the
  // compiler -- possibly in front of the wait)
  bf16x8_t bk[2], bv[2], ndo[2], no[2];
  float nlq = 0.f, nrs = 1.f;
  constexpr bool LS = OT && NP > 1;      // late dK / dV stores (file comment)
  u32x4_t hold[4];                           // LS: the previous item's packed dK | dV pieces
  bf16_t* hold_base = nullptr;               // LS: ... and where they go (null: nothing held)
  auto flush_hold = [&]() __attribute__((always_inline)) {
    int lane_h = lane;                       // (opaque: the addresses are formed here, not kept across the item)
    asm volatile("" : "+v"(lane_h));
    const int key_h = k0 + (lane_h & 15), g_h = lane_h >> 4;
    if (key_h < S) {
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        bf16_t* rp = hold_base + (which + 1) * D + (size_t)key_h * ld + (g_h >> 1) * 8;
        *(u32x4_t*)(rp + (g_h & 1) * 16) = hold[2 * which];
        *(u32x4_t*)(rp + (2 + (g_h & 1)) * 16) = hold[2 * which + 1];
      }
    }
  };
  auto fetch_kv = [&](int it) __attribute__((always_inline)) {
    const size_t r0 = (size_t)(it / H) * S;
    int lane_f = lane;                         // (opaque: the strips' per-lane offsets are recomputed here, not kept across the item)
    asm volatile("" : "+v"(lane_f));
    const int key_f = k0 + (lane_f & 15);
    const uint32_t oq = key_f < S ? (uint32_t)((key_f * ld + (lane_f >> 4) * 8) * 2) : STRIP_OOB;      // row of qkv (the k / v parts through the scalar offset)
    const u32x4_t rq = raw_rsrc(qkv + r0 * ld + (it % H) * HD);
    const uint32_t sk = (uint32_t)(D * 2), sv = (uint32_t)(D * 4);
    bk[0] = strip_ld_na<0>(rq, oq, sk);  bk[1] = strip_ld_na<64>(rq, oq, sk);
    bv[0] = strip_ld_na<0>(rq, oq, sv);  bv[1] = strip_ld_na<64>(rq, oq, sv);
  };
  auto fetch_rows = [&](int it) __attribute__((always_inline)) {
    const int fr = it / H, hd = it % H;
    const size_t r0 = (size_t)fr * S;
    int lane_f = lane;
    asm volatile("" : "+v"(lane_f));
    const int key_f = k0 + (lane_f & 15);
    const bool in = key_f < S;
    if constexpr (!OT) {
      const uint32_t od = in ? (uint32_t)((key_f * D + (lane_f >> 4) * 8) * 2) : STRIP_OOB;       // row of dout / out
      const u32x4_t rd = raw_rsrc(dout + r0 * D + hd * HD), ro = raw_rsrc(out + r0 * D + hd * HD);
      ndo[0] = strip_ld_na<0>(rd, od, 0u); ndo[1] = strip_ld_na<64>(rd, od, 0u);
      no[0] = strip_ld_na<0>(ro, od, 0u);  no[1] = strip_ld_na<64>(ro, od, 0u);
    }
    // per-row scalars of rows past the sequence: the last row's (finite; they only ever meet zeros -- D[q] of such a row is 0 through the zero strips)
    const int kc = in ? key_f : S - 1;
    {
      nlq = dword_ld_na(lse + ((size_t)fr * H + hd) * S, (uint32_t)(kc * 4));              // raw: `* log2(e)` where it is stored to LDS
      if (SCALED) nrs = dword_ld_na(row_scale + 2 * r0, (uint32_t)(kc * 8));
    }
  };
  int item = blockIdx.x;
  if (item < items) {
    const size_t r0 = (size_t)(item / H) * S;
    stage_head_dma(qkv + r0 * ld + (item % H) * HD, ld, S, Qs, KP, wv, NKT, lane);
    stage_head_dma(dout + r0 * D + (item % H) * HD, D, S, dOs, KP, wv, NKT, lane);
    if constexpr (OT) stage_head_dma(out + r0 * D + (item % H) * HD, D, S, Os, NKT * 16, wv, NKT, lane);
    fetch_kv(item);
    fetch_rows(item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  for (; item < items; item += gridDim.x) {
    const int frame = item / H, head = item % H;
    const size_t row0 = (size_t)frame * S;
    bf16_t* dbase = dqkv + row0 * ld + head * HD;
    const bool has_next = item + gridDim.x < items;
    const int nitem = has_next ? item + gridDim.x : item;
    const size_t nr0 = (size_t)(nitem / H) * S;
    __amdgpu_buffer_rsrc_t nrq = __builtin_amdgcn_make_buffer_rsrc((void*)(qkv + nr0 * ld + (nitem % H) * HD), 0, 0x7FFFFFF0u, 0x00020000);
    __amdgpu_buffer_rsrc_t nrdo = __builtin_amdgcn_make_buffer_rsrc((void*)(dout + nr0 * D + (nitem % H) * HD), 0, 0x7FFFFFF0u, 0x00020000);
    __amdgpu_buffer_rsrc_t nro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + nr0 * D + (nitem % H) * HD), 0, 0x7FFFFFF0u, 0x00020000);      // (OT)

    // own strips have landed once everything but the youngest 8 vector-memory operations (the previous item's dK / dV stores) is done;
    // older than the strips are the prefetched Q / dO rows of this item and the dQ stores of the previous one
    // (wide stores: 4)
#define AVT_STRIPS_LANDED(N) do { if constexpr (OT) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bk[0]), "+v"(bk[1]), "+v"(bv[0]), "+v"(bv[1]), "+v"(nlq), "+v"(nrs) :: "memory"); \
    else asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bk[0]), "+v"(bk[1]), "+v"(bv[0]), "+v"(bv[1]), "+v"(ndo[0]), "+v"(ndo[1]), "+v"(no[0]), "+v"(no[1]), "+v"(nlq), "+v"(nrs) :: "memory"); } while (0)
    if (ALL_LIVE && !LS) AVT_STRIPS_LANDED(4);
    else AVT_STRIPS_LANDED(0);               // (LS: the stores are not issued yet -- the strips' requests are the youngest operations)
#undef AVT_STRIPS_LANDED
    auto publish_rows = [&](float dsum) __attribute__((always_inline)) {
      dsum = gsum(dsum) * scale;
      int lane_s = lane;                        // (opaque: these LDS addresses are used once per item)
      asm volatile("" : "+v"(lane_s));
      if (lane_s < 16) { dq_s[k0 + lane_s] = dsum; lse_s[k0 + lane_s] = nlq * LOG2E; if (SCALED) rs_s[k0 + lane_s] = nrs; }         // k0 + 15 < NKT * 16 <= KP; rows past the sequence: 0 / 0
      if (NKT * 16 < KP && wv == 0 && lane_s < KP - NKT * 16) { dq_s[NKT * 16 + lane_s] = 0.f; lse_s[NKT * 16 + lane_s] = 0.f; }
    };
    if constexpr (!OT) {
      float dsum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)ndo[ks][e] * (float)no[ks][e];
      publish_rows(dsum);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // barrier S: (!OT: scalars visible;) every wave is done with the previous item, its share of this item's rows has landed
    stage_head_dma(qkv + row0 * ld + D + head * HD, ld, S, Ks, KP, wv, NKT, lane);       // K tile: first read after barrier 0
    if constexpr (OT) {
      // D of the own strip from the dO and O tiles (B-operand layout = the layout of a row fragment)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      float dsum = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t a = frag_rm(dOs, wv, ks, lane_o), b = frag_rm(Os, wv, ks, lane_o);
#pragma unroll
        for (int e = 0; e < 8; ++e) dsum += (float)a[e] * (float)b[e];
      }
      publish_rows(dsum);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // barrier S2 (scalars visible)
    }
    if (dbias && prev_head >= 0) {            // the previous item's sums, folded in tile order by the column's owner thread
      float* bh = bias_s + prev_head * 128;
      int tid_b = tid;                        // (opaque: the addresses below are formed here, not kept -- spilled -- across the item)
      asm volatile("" : "+v"(tid_b));
      for (int c = tid_b; c < 64; c += nthr) {
        float t = bh[c];
#pragma unroll
        for (int w = 0; w < NKT; ++w) t += stq_s[w * 64 + c];
        bh[c] = t;
        float u = bh[64 + c];
#pragma unroll
        for (int w = 0; w < NP; ++w) u += stv_s[w * 64 + c];
        bh[64 + c] = u;
      }
    }
    prev_head = head;

    f32x4_t adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { adk[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; adv[dt] = adk[dt]; }

    // the item's tail: dK / dV of the wave's strip leave, the next item's per-row scalars (!OT: and dO / O strips) are requested.  (Executing it right after
    // the last chunk's barrier, before that chunk's dQ products, measured no better: file comment.)
    auto item_tail = [&]() __attribute__((always_inline)) {
    // (scaled: this item's scale of the wave's own key rows, back from LDS -- kept in a register across the item it cost 14 spilled registers)
      int lane_t = lane;                          // (opaque: the addresses below are formed here, not kept -- spilled -- across the item)
      asm volatile("" : "+v"(lane_t));
      const int key_t = k0 + (lane_t & 15), g_t = lane_t >> 4;
      float crs = 1.f;
      if (SCALED) crs = rs_s[key_t];
      // next item's per-row scalars (!OT: and dO / O strips): requested after the register-hungry loop, hidden behind the stores and the item's first barrier.
      // Unconditional (re-fetches this item at the end): keeps the counted wait at the loop top exact.  Every earlier place was tried and spills: the top
      // of the last chunk 20-25 registers, behind the last barrier 9-12 (profiles/r05n_attention_boundary.txt)
      fetch_rows(nitem);
      // 16-byte stores (see the forward kernel): lane (i, g) ends up with 8 consecutive columns of block dp + (g & 1); one tensor after the other
      // (eight registers of packed output at a time: the kernel sits at its 128-register limit)
      static_for<0, 2>([&](auto w_) __attribute__((always_inline)) {
        constexpr int which = decltype(w_)::value;                 // 0 = dK -> columns [D, 2D), 1 = dV -> [2D, 3D)
        const f32x4_t (&acc4)[4] = which ? adv : adk;
        u32x4_t sp[2];
#pragma unroll
        for (int dp = 0; dp < 4; dp += 2) {
          u32x2_t a, b;
          a[0] = pack2bf(acc4[dp][0] * crs, acc4[dp][1] * crs); a[1] = pack2bf(acc4[dp][2] * crs, acc4[dp][3] * crs);
          b[0] = pack2bf(acc4[dp + 1][0] * crs, acc4[dp + 1][1] * crs); b[1] = pack2bf(acc4[dp + 1][2] * crs, acc4[dp + 1][3] * crs);
          const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false), r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
          sp[dp >> 1] = (u32x4_t){r0[0], r1[0], r0[1], r1[1]};
        }
        if constexpr (LS) { hold[2 * which] = sp[0]; hold[2 * which + 1] = sp[1]; }
        else {
          bf16_t* tb = dbase + (which + 1) * D;
          if (key_t < S) {
            bf16_t* rp = tb + (size_t)key_t * ld + (g_t >> 1) * 8;
            *(u32x4_t*)(rp + (g_t & 1) * 16) = sp[0];
            *(u32x4_t*)(rp + (2 + (g_t & 1)) * 16) = sp[1];
          }
        }
      });
      if constexpr (LS) hold_base = dbase;
    };
    auto dma_next_rows = [&](int c) __attribute__((always_inline)) {
      // the NEXT item's Q / dO (OT: / O) rows of this chunk (4 + 4 (+ 4) LDS-DMA instructions of 8 rows), spread over the waves
      if (has_next) {
        int lane_r = lane;                      // (opaque: the rows' per-lane offsets are formed here, not kept -- spilled -- across the item)
        asm volatile("" : "+v"(lane_r));
        for (int j = wv; j < (OT ? 12 : 8); j += NKT) {
          if (j < 4) dma_rows8(nrq, ld, S, Qs, 4 * c + j, lane_r);
          else if (j < 8) dma_rows8(nrdo, D, S, dOs, 4 * c + j - 4, lane_r);
          else if (4 * c + j - 8 < 2 * NKT) dma_rows8(nro, D, S, Os, 4 * c + j - 8, lane_r);      // (OT) the O tile ends with the last strip
        }
      }
    };
    static_for<0, NP>([&](auto c_) __attribute__((always_inline)) {
      constexpr int c = decltype(c_)::value;
      // software pipeline: the transposing reads of the dO tile (for dV) are requested first and land under the score / softmax work;
      // those of the Q tile (for dK) are requested before the dV products and land under them
      bf16x8_t tfo[4], tfq[4];
      frag4_tr_na<RM, NP>(tfo, tr0, c);                                     // dO tile, queries of pair c
      f32x4_t pv2[2], ds2[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * c + u;
        pv2[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        ds2[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (qt < NKT) {
          f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = mfma16(frag_rm(Qs, qt, 0, lane), bk[0], s);
          s = mfma16(frag_rm(Qs, qt, 1, lane), bk[1], s);
          dp = mfma16(frag_rm(dOs, qt, 0, lane), bv[0], dp);
          dp = mfma16(frag_rm(dOs, qt, 1, lane), bv[1], dp);
          int g_s = g;                           // (opaque: the scalars' LDS address is formed per chunk, not kept -- spilled -- across the item)
          asm volatile("" : "+v"(g_s));
          const f32x4_t l4 = *(const f32x4_t*)(lse_s + qt * 16 + 4 * g_s);    // pre-scaled by log2(e)
          const f32x4_t d4 = *(const f32x4_t*)(dq_s + qt * 16 + 4 * g_s);     // pre-scaled by `scale`
          const f32x2_t s01 = (f32x2_t){s[0], s[1]} * sl - (f32x2_t){l4[0], l4[1]}, s23 = (f32x2_t){s[2], s[3]} * sl - (f32x2_t){l4[2], l4[3]};
          const f32x2_t e01 = (f32x2_t){dp[0], dp[1]} * scale - (f32x2_t){d4[0], d4[1]}, e23 = (f32x2_t){dp[2], dp[3]} * scale - (f32x2_t){d4[2], d4[3]};
          const f32x2_t p01 = (f32x2_t){__builtin_amdgcn_exp2f(s01[0]), __builtin_amdgcn_exp2f(s01[1])};
          const f32x2_t p23 = (f32x2_t){__builtin_amdgcn_exp2f(s23[0]), __builtin_amdgcn_exp2f(s23[1])};
          const f32x2_t d01 = p01 * e01, d23 = p23 * e23;
          pv2[u] = (f32x4_t){p01[0], p01[1], p23[0], p23[1]};
          ds2[u] = (f32x4_t){d01[0], d01[1], d23[0], d23[1]};
        }
      }
      // (the last chunk's score products were the last readers of this item's K / V strips)
      if constexpr (c == NP - 1) fetch_kv(nitem);
      const bf16x8_t bp = pack_pair(pv2[0], pv2[1]);
      union { bf16x8_t v; uint32_t w[4]; } bd;
      bd.v = pack_pair(ds2[0], ds2[1]);
      // dS of (own keys) x (the chunk's 32 queries) -> buffer c & 1: two 8-byte stores per lane (queries 4g..4g+3 of each tile); the buffer's
      // previous contents (chunk c - 2) were consumed before barrier c - 1
      *(u32x2_t*)(dswr + (c & 1) * DSB) = (u32x2_t){bd.w[0], bd.w[1]};
      *(u32x2_t*)(dswr + (c & 1) * DSB + 32) = (u32x2_t){bd.w[2], bd.w[3]};
      asm volatile("" ::: "memory");                                          // (the stores above are queued before the reads below)
      bool held = false;
      if constexpr (LS && c == 0) { held = hold_base != nullptr; if (held) flush_hold(); }        // the previous item's dK / dV (uniform branch)
      frag4_tr_na<0, NP>(tfq, tr0, c);                                       // Q tile: 8 reads, the youngest LDS operations of this wave
      asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(tfo[0]), "+v"(tfo[1]), "+v"(tfo[2]), "+v"(tfo[3]));   // everything older has returned: the dO fragments
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) adv[dt] = mfma16(tfo[dt], bp, adv[dt]);
      if (dbias && wv == c) {
        // colsum(dV) = sum_k sum_q P[q,k] dO[q] = sum_q dO[q] (every row of P sums to one): wave c takes the 32 queries of pair c from the
        // fragments it has just read, as one more product with an all-ones B operand, and parks the 64 sums in its staging row
        union { bf16x8_t v; uint32_t w[4]; } ones;
        ones.w[0] = ones.w[1] = ones.w[2] = ones.w[3] = 0x3F803F80u;
        f32x4_t cs[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) cs[dt] = mfma16(tfo[dt], ones.v, (f32x4_t){0.f, 0.f, 0.f, 0.f});
        // (the 11 wait states an 8-pass MFMA result needs before a memory instruction reads it, explicit and tied to the registers)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(cs[0]), "+v"(cs[1]), "+v"(cs[2]), "+v"(cs[3]));
        if (i16 == 0) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) *(f32x4_t*)(stv_s + c * 64 + dt * 16 + 4 * g) = cs[dt];
        }
      }
      lgkm_wait4(tfq[0], tfq[1], tfq[2], tfq[3]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) adk[dt] = mfma16(tfq[dt], bd.v, adk[dt]);
      if (c == 0) {                                   // this wave's share of the K tile has landed
        if (LS && ALL_LIVE && held) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");           // (all but the four dK / dV stores just issued)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // barrier c: dS chunk complete; the chunk's Q / dO rows are free
      (void)item_tail;      // (keeps `item_tail` captured by this lambda as it was while the tail could also run here: without the capture hipcc 7.2 lays the
                            //  closure out differently and the kernel comes out with a spilled vector register -- compared on the ISA, round 6)
      // ---- dQ of the chunk's two query tiles: four (tile, half) products, one wave each ----
      static_for<0, 4>([&](auto hh_) __attribute__((always_inline)) {
        constexpr int h = 4 * c + decltype(hh_)::value;
        if constexpr (h < 2 * NKT) {
          if (wv == h % NKT) {
            constexpr int qt = h >> 1, half = h & 1, u = qt & 1;
            f32x4_t acc[2];
            acc[0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
            // fragments of key pair t + 1 are requested before the products of pair t (two register sets)
            bf16x8_t bb[2], ka[2], kb[2];
            auto rd = [&](auto t_) __attribute__((always_inline)) {
              constexpr int t = decltype(t_)::value;
              constexpr int o = (c & 1) * DSB + 32 * t * DSP + 32 * u;
              const u32x2_t lo = ds_read_tr_na<o>(dsrd), hi = ds_read_tr_na<o + 16 * DSP>(dsrd);     // keys 32t + 4g + j | 32t + 16 + 4g + j
              bb[t & 1] = tr_join(lo, hi);
              ka[t & 1] = frag_tr_na<0, t>(trK[2 * half]);
              kb[t & 1] = frag_tr_na<0, t>(trK[2 * half + 1]);
            };
            rd(std::integral_constant<int, 0>{});
            static_for<0, NP>([&](auto t_) __attribute__((always_inline)) {
              constexpr int t = decltype(t_)::value;
              if constexpr (t + 1 < NP) {
                rd(std::integral_constant<int, t + 1>{});
                asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bb[t & 1]), "+v"(ka[t & 1]), "+v"(kb[t & 1]));     // the 6 reads just queued may still be out
              } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb[t & 1]), "+v"(ka[t & 1]), "+v"(kb[t & 1]));
              }
              acc[0] = mfma16(ka[t & 1], bb[t & 1], acc[0]);
              acc[1] = mfma16(kb[t & 1], bb[t & 1], acc[1]);
            });
            int lane_d = lane;                       // (opaque: the store addresses are recomputed here instead of being kept -- spilled -- across the item)
            asm volatile("" : "+v"(lane_d));
            const int q = qt * 16 + (lane_d & 15);
            {
              u32x2_t a, b;
              float rq = 1.f;
              if (SCALED) {       // the query rows' scale, read here and now (inline assembly: hoisted above the products by the compiler it cost 13 spilled registers)
                const uint32_t ra_ = lds_addr32((const char*)rs_s) + (uint32_t)((qt * 16 + (lane_d & 15)) * 4);
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rq) : "v"(ra_) : "memory");
              }
              a[0] = pack2bf(acc[0][0] * rq, acc[0][1] * rq); a[1] = pack2bf(acc[0][2] * rq, acc[0][3] * rq);
              b[0] = pack2bf(acc[1][0] * rq, acc[1][1] * rq); b[1] = pack2bf(acc[1][2] * rq, acc[1][3] * rq);
              const auto r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false), r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
              const int gd = lane_d >> 4;
              if (q < S) *(u32x4_t*)(dbase + (size_t)q * ld + (2 * half + (gd & 1)) * 16 + (gd >> 1) * 8) = (u32x4_t){r0[0], r1[0], r0[1], r1[1]};
            }
            if (dbias) {
              // q part of the qkv-bias gradient: sums over the tile's 16 queries = the 16 lanes of a DPP row (rows past the sequence are
              // exactly zero).  The accumulators come straight out of the matrix pipe into hand-written DPP adds: the wait states are
              // tied to the registers so that no MFMA can be scheduled below them
              asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
              lsum16x4(acc[0]); lsum16x4(acc[1]);
              if (i16 == 0) {
                *(f32x4_t*)(stq_s + qt * 64 + (2 * half) * 16 + 4 * g) = acc[0];
                *(f32x4_t*)(stq_s + qt * 64 + (2 * half + 1) * 16 + 4 * g) = acc[1];
              }
            }
          }
        }
      });
      dma_next_rows(c);
    });

    item_tail();
  }
  // the last item re-requested its own strips (that keeps the counted waits exact): retire those requests before anything below reuses their registers --
  // the compiler considers them dead here, and a load landing in a register that meanwhile holds an address is a wild store (tools/isa_async_check.py found it)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(bk[0]), "+v"(bk[1]), "+v"(bv[0]), "+v"(bv[1]), "+v"(ndo[0]), "+v"(ndo[1]), "+v"(no[0]), "+v"(no[1]), "+v"(nlq), "+v"(nrs) :: "memory");
  if constexpr (LS) { if (hold_base != nullptr) flush_hold(); }
  if (dbias) {
    __syncthreads();
    if (prev_head >= 0) {
      float* bh = bias_s + prev_head * 128;
      for (int c = tid; c < 64; c += nthr) {
        float t = bh[c];
#pragma unroll
        for (int w = 0; w < NKT; ++w) t += stq_s[w * 64 + c];
        bh[c] = t;
        float u = bh[64 + c];
#pragma unroll
        for (int w = 0; w < NP; ++w) u += stv_s[w * 64 + c];
        bh[64 + c] = u;
      }
    }
    __syncthreads();
    for (int i = tid; i < H * 192; i += nthr) {
      const int hh = i / 192, c = i % 192;
      const int pt = c >> 6;                                              // 0 = q, 1 = k (zero), 2 = v
      const float v = pt == 1 ? 0.f : bias_s[hh * 128 + (pt >> 1) * 64 + (c & 63)];
      const int o = (c >> 6) * D + hh * HD + (c & 63);
      if (part) part[(size_t)blockIdx.x * (3 * D) + o] = v;
      else if (v != 0.f) unsafeAtomicAdd(&dbias[o], v);
    }
  }
}

int pick_nkt(int S) { int n = (S + 15) / 16; if (n <= 1) return 1; if (n <= 2) return 2; if (n <= 4) return 4; if (n <= 8) return 8; return 13; }

template <int NKT> size_t fwd_smem() { constexpr int NP = (NKT + 1) / 2; return (size_t)4 * NP * 32 * 128; }     // K, V x 2 buffers

constexpr size_t BWD1_LAB_SMEM = 0;
template <int NKT> size_t bwd1_smem(int H) { constexpr int NP = (NKT + 1) / 2, KP = NP * 32; return (size_t)3 * KP * 128 + (size_t)2 * KP * 72 + (size_t)(3 * KP + H * 128 + NKT * 64 + NP * 64) * 4 + BWD1_LAB_SMEM; }

template <int NKT>
int launch_fwd(const bf16_t* qkv, bf16_t* out, float* lse, int frames, int S, int H, float scale, hipStream_t s) {
  size_t sm = fwd_smem<NKT>();
  (void)hipFuncSetAttribute((const void*)vit_attn_fwd_kernel<NKT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  (void)hipFuncSetAttribute((const void*)vit_attn_fwd_kernel<NKT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  const int items = frames * H;
  const bool all_live = S > (NKT - 1) * 16;
  const int per_cu = (int)((160 * 1024) / sm) < 1 ? 1 : (int)((160 * 1024) / sm);
  int grid = 256 * (per_cu > 8 ? 8 : per_cu);
  if (grid > items) grid = items;
  if (all_live) hipLaunchKernelGGL((vit_attn_fwd_kernel<NKT, true>), dim3(grid), dim3(64 * NKT), sm, s, qkv, out, lse, S, H, items, scale);
  else hipLaunchKernelGGL((vit_attn_fwd_kernel<NKT, false>), dim3(grid), dim3(64 * NKT), sm, s, qkv, out, lse, S, H, items, scale);
  return 0;
}
template <int NKT>
int launch_bwd(const bf16_t* qkv, const bf16_t* out, const bf16_t* dout, const float* lse, bf16_t* dqkv, float* dbias,
               float* part, size_t part_bytes, int frames, int S, int H, float scale, hipStream_t s, const float* row_scale = nullptr) {
  {
    size_t sm1 = bwd1_smem<NKT>(H);
    if (sm1 > 160 * 1024) { avt_set_error("avt_vit_attn_bwd: H = %d needs more LDS than a CU has", H); return -1; }
    // the O tile (kernel comment) where it fits: H <= 21 at NKT = 13
    const bool ot = sm1 + (size_t)NKT * 16 * 128 <= 160 * 1024;
    if (ot) sm1 += (size_t)NKT * 16 * 128;
    const bool live = S > (NKT - 1) * 16;
    const int items1 = frames * H;
    const int pc = (int)((160 * 1024) / sm1) < 1 ? 1 : (int)((160 * 1024) / sm1);
    int grid1 = 256 * (pc > 8 ? 8 : pc);
    if (grid1 > items1) grid1 = items1;
    if (!dbias) part = nullptr;
    if (part && part_bytes < (size_t)grid1 * 3 * H * HD * 4) { avt_set_error("avt_vit_attn_bwd: partials workspace too small"); return -1; }
#define AVT_BWD1_(LIVE, SC, OTV) do { (void)hipFuncSetAttribute((const void*)vit_attn_bwd1_kernel<NKT, LIVE, SC, OTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1); \
    hipLaunchKernelGGL((vit_attn_bwd1_kernel<NKT, LIVE, SC, OTV>), dim3(grid1), dim3(64 * NKT), sm1, s, qkv, out, dout, lse, dqkv, dbias, part, S, H, items1, scale, row_scale); } while (0)
#define AVT_BWD1(LIVE, SC) do { if (ot) AVT_BWD1_(LIVE, SC, true); else AVT_BWD1_(LIVE, SC, false); } while (0)
    if (row_scale) { if (live) AVT_BWD1(true, true); else AVT_BWD1(false, true); }
    else { if (live) AVT_BWD1(true, false); else AVT_BWD1(false, false); }
#undef AVT_BWD1_
#undef AVT_BWD1
    if (part) { float* outs[1] = {dbias}; return avt_reduce_partials(part, grid1, 3L * H * HD, outs, 1, s); }
    return 0;
  }
}

}  // namespace

extern "C" int avt_vit_attn_fwd(const void* qkv, void* out, float* lse, int frames, int S, int H, int head_dim, float scale, void* stream) {
  AVT_CHECK(qkv && out && lse, "avt_vit_attn_fwd: null argument");
  AVT_CHECK(head_dim == 64, "avt_vit_attn_fwd: head_dim must be 64 (got %d)", head_dim);
  AVT_CHECK(S >= 1 && S <= 208, "avt_vit_attn_fwd: S must be in [1, 208] (got %d)", S);
  AVT_CHECK(frames > 0 && H > 0 && aligned16(qkv) && aligned16(out), "avt_vit_attn_fwd: bad shape or alignment");
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* q = (const bf16_t*)qkv; bf16_t* o = (bf16_t*)out;
  switch (pick_nkt(S)) {
    case 1: launch_fwd<1>(q, o, lse, frames, S, H, scale, s); break;
    case 2: launch_fwd<2>(q, o, lse, frames, S, H, scale, s); break;
    case 4: launch_fwd<4>(q, o, lse, frames, S, H, scale, s); break;
    case 8: launch_fwd<8>(q, o, lse, frames, S, H, scale, s); break;
    default: launch_fwd<13>(q, o, lse, frames, S, H, scale, s); break;
  }
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t avt_vit_attn_bwd_workspace_bytes(int frames, int S, int H) {
  (void)frames; (void)S;
  return (size_t)2048 * 3 * (size_t)H * HD * 4;                   // at most 256 x 8 persistent workgroups
}
static int vit_attn_bwd_impl(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias,
                             int frames, int S, int H, int head_dim, float scale, float* part, size_t part_bytes, const float* row_scale, void* stream) {
  AVT_CHECK(qkv && out && dout && lse && dqkv, "avt_vit_attn_bwd: null argument");
  AVT_CHECK(head_dim == 64, "avt_vit_attn_bwd: head_dim must be 64 (got %d)", head_dim);
  AVT_CHECK(S >= 1 && S <= 208, "avt_vit_attn_bwd: S must be in [1, 208] (got %d)", S);
  AVT_CHECK(frames > 0 && H > 0 && aligned16(qkv) && aligned16(out) && aligned16(dout) && aligned16(dqkv), "avt_vit_attn_bwd: bad shape or alignment");
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* q = (const bf16_t*)qkv; const bf16_t* o = (const bf16_t*)out; const bf16_t* d = (const bf16_t*)dout; bf16_t* dq = (bf16_t*)dqkv;
  int rc = 0;
  switch (pick_nkt(S)) {
    case 1: rc = launch_bwd<1>(q, o, d, lse, dq, dbias, part, part_bytes, frames, S, H, scale, s, row_scale); break;
    case 2: rc = launch_bwd<2>(q, o, d, lse, dq, dbias, part, part_bytes, frames, S, H, scale, s, row_scale); break;
    case 4: rc = launch_bwd<4>(q, o, d, lse, dq, dbias, part, part_bytes, frames, S, H, scale, s, row_scale); break;
    case 8: rc = launch_bwd<8>(q, o, d, lse, dq, dbias, part, part_bytes, frames, S, H, scale, s, row_scale); break;
    default: rc = launch_bwd<13>(q, o, d, lse, dq, dbias, part, part_bytes, frames, S, H, scale, s, row_scale); break;
  }
  if (rc) return rc;
  AVT_LAUNCH_CHECK();
  return 0;
}

extern "C" int avt_vit_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias,
                                int frames, int S, int H, int head_dim, float scale, float* part, size_t part_bytes, void* stream) {
  return vit_attn_bwd_impl(qkv, out, dout, lse, dqkv, dbias, frames, S, H, head_dim, scale, part, part_bytes, nullptr, stream);
}
extern "C" int avt_vit_attn_bwd_scaled(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias,
                                       int frames, int S, int H, int head_dim, float scale, float* part, size_t part_bytes,
                                       const float* row_stat, void* stream) {
  AVT_CHECK(row_stat && (((uintptr_t)row_stat) & 7) == 0, "avt_vit_attn_bwd_scaled: row_stat must be an 8-byte aligned [frames * S][2] fp32 array");
  return vit_attn_bwd_impl(qkv, out, dout, lse, dqkv, dbias, frames, S, H, head_dim, scale, part, part_bytes, row_stat, stream);
}
