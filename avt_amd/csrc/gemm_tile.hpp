// Shared pieces of the bf16 MFMA GEMM kernels (gemm.hip, gemm_persist.hip): parameter block, LDS-DMA operand loaders,
// fragment reads, GELU table, epilogue building blocks.  Everything but the parameter block lives in an anonymous namespace: each
// translation unit that includes this file gets its own copy (device code is not linked across objects).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "../../include/avt_hip.h"

struct GemmParams {
  const bf16_t* A; const bf16_t* B; void* C; bf16_t* C2;
  const float* bias; const bf16_t* res; const bf16_t* aux; float* colsum;
  float* colsum_part;                      // [row slots][N] partial column sums (one row per wave row of the grid) instead of atomics
  int M, N, K;
  int lda, ldb, ldc, ldc2, ldres, ldaux;
  int res_period;
  int act;          // 0 none | 1 gelu_erf | 2 gelu_tanh (C2 = derivative) | 3 *= aux
  int out_f32;
  int splitk;
  int tiles_m, tiles_n;
  int strip_w;                   // 8-phase kernel, activation epilogue: walk the tiles in column strips of this many tiles (0 = row-major)
  size_t ws_bytes;
  float* ws;                     // EPI 2: split-K partial slabs, [splitk * tiles][BM * BN] fp32 in accumulator order
  uint32_t a_bytes, b_bytes;     // buffer-descriptor bounds
  uint32_t drop_thresh; float drop_scale; uint64_t drop_seed;
  int wide_ok;                   // all epilogue leading dims are multiples of 8 -> 16-byte accesses allowed
  // LayerNorm folded into the GEMMs around it (avt_gemm_ln_bf16, include/avt_hip.h):
  //   ln_c != NULL ("fold"): A holds the UN-normalised rows x, B = gamma o W, and v = rstd[m] * acc + (bias[n] - mean[m] * rstd[m] * ln_c[n]) with
  //                          ln_stat[m] = {rstd, -mean * rstd} -- the LayerNorm of the rows without a normalised copy of them;
  //   ln_c == NULL, ln_stat != NULL ("scale", act 3 only): the output rows are multiplied by ln_stat[m][0] (= rstd) on their way out and the column
  //                          sums are taken over the UNscaled values (weights ln_stat[m][1] = 1 / rstd): dY' = rstd o dY for the folded backward;
  //   stat_part != NULL: per-row partial (sum, sum of squares) of the output over each 32-column slot, [ceil(N / 64)][M][2 slots][2] fp32 -- the next LayerNorm's statistics.
  const float* ln_stat; const float* ln_c; float* stat_part;
  // fragment-major second output / second operand (ldc2 == 0 / ldaux == 0 in the C ABI; gemm_persist.hip: the persistent kernel only)
  int c2_frag, aux_frag;
  // weight-gradient slabs (EPI 2): walk a split's tiles column-major (the output has more column tiles than row tiles), so that the contiguous
  // range of the walk an XCD owns covers few operand panels either way (accum_slab; profiles/r06h_wgrad_xcd.txt)
  int tile_cm;
  // EPI 2 (deterministic weight gradients): C = result instead of C += result (avt_gemm_assign_bf16: the caller knows C holds zeros -- the fused optimizer
  // has just re-zeroed the gradient buffer -- so reading it back is 4 bytes per weight for nothing)
  int c_assign;
};
// gemm_persist.hip: the persistent form of the 8-phase kernel (one workgroup per CU walks a queue of output tiles and keeps the
// next tile's first operand half-tiles in flight while it converts and stores the current one).  Returns 1 = launched,
// 0 = this shape / epilogue is not covered (the caller falls back to gemm_8p_kernel), < 0 = error (avt_set_error called).
// `kinds` = bit mask of the epilogue kinds (EPK 0..3) the caller allows (the product allows all; the lab build's A/B switch);
// `force`: also take reductions longer than the range where the persistent form measured faster (tile 809).
int avt_gemm_persist(GemmParams& p, int kinds, bool force, hipStream_t s);

namespace {

// Instrumentation and experiment switches exist only in the lab build (make lab -> libavt_hip_lab.so, used by tools/):
// the product library reads no environment variable and takes no pointer from anywhere but its arguments.

constexpr int BK64 = 64;
// (cache policy of the operand streams and of the epilogue's second operand: default -- nt on the A stream costs 0.3-0.6 %, on the second operand it is
// neutral; profiles/r04_cache_policy.txt)

// XCD-aware bijective remap of the linear block id: XCD x (= id % 8 by dispatch order) owns a contiguous
// range of logical blocks, ordered (split, tile row, tile column), so tiles sharing an A row-panel sit behind the same L2
// and -- for split-K weight gradients -- an XCD works on one or two K ranges only, instead of pulling every K range of
// the shared B operand through each of the eight L2s (fc1 wgrad: 2.65 GB fetched for 0.97 GB of operands before).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7;
  int xcd = bid & 7, idx = bid >> 3;
  int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}
// Split-K slab of this workgroup = split * tiles + (row-major tile index): what splitk_reduce_kernel expects.  The WALK (which slab the workgroup at
// position `xcd_remap(blockIdx.x)` takes) runs through a split's tiles along the SHORTER side of the tile grid first: an XCD owns a contiguous range of
// ~32 positions, and its L2 serves every operand panel of that range once -- a 3 x 12 grid walked row-major makes such a range touch 3 + 12 panels of one
// split and 1 + 4 .. 2 + 12 of the next, walked column-major 3 + 11 and 2 + 3 (fc2's weight gradient at 256 clips: 6.42 -> 4.8 GB of fabric reads per launch
// for 3.87 GB of operands, profiles/r06h_wgrad_xcd.txt).
__device__ __forceinline__ int accum_slab(const GemmParams& p) {
  const int ntile = p.tiles_m * p.tiles_n;
  const int pos = xcd_remap(blockIdx.x, ntile * p.splitk);
  if (!p.tile_cm) return pos;
  const int split = pos / ntile, u = pos - split * ntile, tn = u / p.tiles_m;
  return split * ntile + (u - tn * p.tiles_m) * p.tiles_n + tn;
}

// One MFMA step; the operand order decides whether a lane ends up holding a column or a row of the output block:
// EPI 0 (activation epilogue): B first, so block (i,j) comes out TRANSPOSED -- lane l holds output row i*32 + (l&31) and its
//   16 registers are four groups q of 4 CONSECUTIVE columns j*32 + 8q + 4(l>>5) + (r&3): elementwise work, bf16 packing and
//   the LDS hand-off all work on 8-/16-byte units instead of single floats.
// EPI 1 (atomic accumulate): A first, lane l holds column j*32 + (l&31), so a wave's atomics hit 128 consecutive bytes.
template <int EPI>
__device__ __forceinline__ f32x16_t mma(bf16x8_t a, bf16x8_t b, f32x16_t c) {
  return EPI == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- GELU by table (fc1 forward: GELU + GELU' outputs) --------------------------------------------------------------------------
// The erf polynomial + exponential of gelu_erf_both4 is 36 packed / conversion instructions + 4 v_exp per 4 elements, and the
// epilogue of these launches is bound by the vector-ALU issue rate (profiles/r03x_pmc_sq.txt: 353 M non-MFMA vector instructions per
// launch = 0.26 of the SIMD-cycles next to 0.43 of MFMA).  The table form rounds the biased pre-activation to bf16 (one
// v_cvt_pk_bf16_f32 per pair -- the same rounding the stored activation would get one step later) and reads {GELU, GELU'} as a packed
// bf16 pair from a 24-KB LDS table indexed by the bf16 bits: 24 exponents (2^-16 <= |x| < 2^8) x 128 mantissas x sign, generated by
// tools/gen/gelu_table.py from the exact erf form in double precision.  11 vector instructions + 2 ds_read_b32 per PAIR of elements.
// Outside the table: |x| < 2^-16 takes the entry of 2^-16 (|error| <= 7.7e-6 on GELU, GELU' = 0.5 is exact to bf16); |x| >= 2^8 (or NaN)
// is handled exactly by a fix-up pass that a block only enters when one of its values is that large (tracked with one v_pk_max_u16 per pair).
constexpr int GELU_TAB_ELO = 111, GELU_TAB_NEXP = 24, GELU_TAB_NT = GELU_TAB_NEXP * 128;
constexpr int GELU_TAB_BYTES = 2 * GELU_TAB_NT * 4;                       // 24576
constexpr float GELU_TAB_TOP = 256.f;                                     // = 2^(GELU_TAB_ELO + GELU_TAB_NEXP - 127): first magnitude above the table
static_assert(GELU_TAB_ELO + GELU_TAB_NEXP - 127 == 8, "GELU_TAB_TOP");
__device__ const uint32_t g_gelu_tab[2 * GELU_TAB_NT] = {
#include "gelu_table.inc"
};
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;
// the packed bf16 pair w = {x0, x1} -> the byte offsets of its two table entries, packed as 16-bit halves; mx tracks max |bits|
__device__ __forceinline__ uint32_t gelu_tab_offsets(uint32_t w, u16x2_t& mx) {
  constexpr unsigned short LO = GELU_TAB_ELO << 7, SPAN = (GELU_TAB_NEXP << 7) - 1;
  const u16x2_t m = __builtin_bit_cast(u16x2_t, w & 0x7fff7fffu);
  mx = __builtin_elementwise_max(mx, m);
  // byte offset ((clamp(m, LO, HI) - LO) * 2 + sign) * 4 for both halves at once: a saturating subtract does the lower clamp, one min the upper,
  // the sign bits go to bits 2 / 18 with ONE 32-bit shift (what it drags across the halves is masked away) and are OR-ed in by v_and_or_b32:
  // 7 vector instructions per pair of elements (9 before)
  const u16x2_t d = __builtin_elementwise_min(__builtin_elementwise_sub_sat(m, (u16x2_t){LO, LO}), (u16x2_t){SPAN, SPAN});
  const u16x2_t i8 = d << (u16x2_t){3, 3};
  return ((w >> 13) & 0x00040004u) | __builtin_bit_cast(uint32_t, i8);
}
// exact values for a pre-activation the table does not cover from above (|x| >= 2^8, inf, NaN): GELU = x | -0 (|GELU(x)| < 1e-300 there),
// GELU' = 1 | 0
__device__ __forceinline__ void gelu_big(float x, bf16_t& h, bf16_t& d) {
  if (x != x) { h = 0x7fc0; d = 0x7fc0; return; }
  h = x > 0.f ? f2bf(x) : (bf16_t)0x8000; d = x > 0.f ? (bf16_t)0x3f80 : (bf16_t)0;
}
// The same look-up from the table in GLOBAL memory, element by element, for every other place an erf GELU is evaluated (small-tile
// kernels, epilogues with further terms): all erf-GELU epilogues agree bit for bit, whatever tile a shape is routed to.
__device__ __forceinline__ void gelu_tab_scalar(float x, float& y, float& dy) {
  u16x2_t mx = {0, 0};
  const uint32_t off = gelu_tab_offsets(pack2bf(x, x), mx) & 0xffffu;
  const uint32_t e = *(const uint32_t*)((const char*)g_gelu_tab + off);
  y = bflo(e); dy = bfhi(e);
  const float xr = bf2f(f2bf(x));                                            // the bf16-rounded pre-activation, as in the look-up
  if (!(fabsf(xr) < GELU_TAB_TOP)) { bf16_t hb, db; gelu_big(xr, hb, db); y = bf2f(hb); dy = bf2f(db); }
}
__device__ __forceinline__ void gelu_tab_both4(f32x2_t& x0, f32x2_t& x1, f32x2_t& d0, f32x2_t& d1) {
  float y, d;
  gelu_tab_scalar(x0[0], y, d); x0[0] = y; d0[0] = d;
  gelu_tab_scalar(x0[1], y, d); x0[1] = y; d0[1] = d;
  gelu_tab_scalar(x1[0], y, d); x1[0] = y; d1[0] = d;
  gelu_tab_scalar(x1[1], y, d); x1[1] = y; d1[1] = d;
}

// ---- operand tile loaders (LDS-DMA, swizzle on the source address) ------------------------------------
// k-major operand: LDS tile [BR][BK] bf16 (BK*2-byte rows).  The 16-B chunk c of row r lives at chunk
// c ^ ((r>>1)&7) for BK=64 (128-B rows) and c ^ ((r>>2)&3) for BK=32 (64-B rows): the 16 rows a ds_read_b128 lane
// group touches then land on 16 distinct 16-B slots of the 256-B bank row.
template <int BR>
__device__ __forceinline__ int kstrided_swz_fwd(int r) { return BR >= 128 ? ((r & 3) << 2) : (BR >= 64 ? (((r >> 1) & 1) << 2) : 0); }
template <int BK>
__device__ __forceinline__ int kmajor_swz(int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kmajor_part(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int row0, int k0, int ld,
                                                  int K, int wave, int lane, int j) {
  constexpr int CPR = BK / 8;
  constexpr int RPI = 64 / CPR;
  int r = j * (NW * RPI) + wave * RPI + lane / CPR;
  int c = (lane % CPR) ^ kmajor_swz<BK>(r);
  int kcol = k0 + c * 8;
  uint32_t off = (uint32_t)(((size_t)(row0 + r) * (size_t)ld + (size_t)kcol) * 2);
  if (kcol >= K) off = 0xFFFFFFF0u;
  char* dst = lds_tile + (j * (NW * RPI) + wave * RPI) * (BK * 2);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
}
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kstrided_part(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int col0, int k0, int ld,
                                                    int ncols, int wave, int lane, int j) {
  constexpr int CPR = BR / 8;
  constexpr int RPI = 64 / CPR;
  int r = j * NW * RPI + wave * RPI + lane / CPR;
  int c = (lane % CPR) ^ kstrided_swz_fwd<BR>(r);
  int col = col0 + c * 8;
  uint32_t off = (uint32_t)(((size_t)(k0 + r) * (size_t)ld + (size_t)col) * 2);
  if (col >= ncols) off = 0xFFFFFFF0u;
  char* dst = lds_tile + (j * NW * RPI + wave * RPI) * (BR * 2);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
}
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kmajor(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int row0, int k0, int ld,
                                             int K, int wave, int lane) {
  constexpr int CPR = BK / 8;            // 16-B chunks per row
  constexpr int RPI = 64 / CPR;          // rows per wave instruction
#pragma unroll
  for (int j = 0; j < BR / (NW * RPI); ++j) {
    int r = j * (NW * RPI) + wave * RPI + lane / CPR;
    int c = (lane % CPR) ^ kmajor_swz<BK>(r);
    int kcol = k0 + c * 8;
    uint32_t off = (uint32_t)(((size_t)(row0 + r) * (size_t)ld + (size_t)kcol) * 2);
    if (kcol >= K) off = 0xFFFFFFF0u;                     // forces the bounds check -> zeros
    char* dst = lds_tile + (j * (NW * RPI) + wave * RPI) * (BK * 2);   // wave-uniform base; lane l lands at +16*l
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
  }
}
// reduction-index-as-row operand: LDS tile [BK][BR] bf16; chunk swizzle keeps the 4 rows of a tr-read on
// distinct 64-B bank segments (BR>=128: c ^ ((r&3)<<2); BR=64: c ^ (((r>>1)&1)<<2); BR=32: 64-byte rows, the four rows of a read
// already lie 64 B apart -- no swizzle).
template <int BR>
__device__ __forceinline__ int kstrided_swz(int r) { return BR >= 128 ? ((r & 3) << 2) : (BR >= 64 ? (((r >> 1) & 1) << 2) : 0); }
template <int BR, int NW, int BK>
__device__ __forceinline__ void stage_kstrided(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int col0, int k0, int ld,
                                               int ncols, int wave, int lane) {
  constexpr int CPR = BR / 8;            // 16-B chunks per row
  constexpr int RPI = 64 / CPR;          // rows per wave instruction
#pragma unroll
  for (int j = 0; j < BK / (NW * RPI); ++j) {
    int r = j * NW * RPI + wave * RPI + lane / CPR;
    int c = (lane % CPR) ^ kstrided_swz<BR>(r);
    int col = col0 + c * 8;
    uint32_t off = (uint32_t)(((size_t)(k0 + r) * (size_t)ld + (size_t)col) * 2);
    if (col >= ncols) off = 0xFFFFFFF0u;
    char* dst = lds_tile + (j * NW * RPI + wave * RPI) * (BR * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, AVT_LDS_PTR(dst), 16, off, 0, 0, 0);
  }
}

// ---- fragment reads --------------------------------------------------------------------------------------
// Both forms give lane l the 8 values k = ks*16 + (l>>5)*8 + e, e = 0..7, of operand row (tile*32 + (l&31)).
template <int BK>
__device__ __forceinline__ bf16x8_t frag_kmajor(const char* lds_tile, int tile, int ks, int lane) {
  int r = tile * 32 + (lane & 31);
  int c = (ks * 2 + (lane >> 5)) ^ kmajor_swz<BK>(r);
  return *(const bf16x8_t*)(lds_tile + r * (BK * 2) + c * 16);
}
template <int BR>
__device__ __forceinline__ bf16x8_t frag_kstrided(const char* lds_tile, int tile, int ks, int lane) {
  int g = lane >> 4, i16 = lane & 15;
  int col = tile * 32 + (g & 1) * 16 + (i16 & 3) * 4;
  int rbase = ks * 16 + (g >> 1) * 8 + (i16 >> 2);
  union { bf16x8_t v; s16x4_t h[2]; } u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    int r = rbase + h * 4;
    int c = (col >> 3) ^ kstrided_swz<BR>(r);
    const char* p = lds_tile + r * (BR * 2) + c * 16 + (col & 7) * 2;
    u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p));
  }
  return u.v;
}

// frag_kstrided through inline assembly (see ds_read_tr_na in common.hpp): no compiler-placed s_waitcnt vmcnt(0); the caller
// waits (frag_wait) before the first MFMA that consumes the fragment.  Rows r and r + 4 share the swizzle: one address, two immediates.
template <int BR>
__device__ __forceinline__ bf16x8_t frag_kstrided_na(const char* lds_tile, int tile, int ks, int lane) {
  const int g = lane >> 4, i16 = lane & 15;
  const int col = tile * 32 + (g & 1) * 16 + (i16 & 3) * 4;
  const int r = ks * 16 + (g >> 1) * 8 + (i16 >> 2);
  const int c = (col >> 3) ^ kstrided_swz<BR>(r);
  const uint32_t a = lds_addr32(lds_tile + r * (BR * 2) + c * 16 + (col & 7) * 2);
  const u32x2_t lo = ds_read_tr_na<0>(a), hi = ds_read_tr_na<4 * BR * 2>(a);
  return tr_join(lo, hi);
}
// wait until at most N of this wave's LDS operations are outstanding, then pass the fragments through an empty statement so
// that no MFMA reading them can be scheduled above the wait
template <int N, int TM, int TN>
__device__ __forceinline__ void frag_wait(bf16x8_t (&a)[TM], bf16x8_t (&b)[TN]) {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
  for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
  for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(b[j]));
}
// The same transposing read issued through inline assembly, for kernels that place their own s_waitcnt.  hipcc cannot tell that
// a __builtin_amdgcn_ds_read_tr16_b64 does not alias the LDS-DMA (buffer_load ... lds) transfers still in flight and puts an
// s_waitcnt vmcnt(0) in front of every group of them: the whole ring drains before each fragment read and the 1.5-K-tile
// prefetch of the 8-phase kernel degenerates to none (found in round 3: every k-strided operand -- all weight gradients, the
// proj data gradient -- had been running like that; plain ds_read_b128 loads are not affected).  `addr` = the lane's LDS byte
// address, OFF = compile-time offset (slot, k-step, half).  The result may only be used after an explicit s_waitcnt lgkmcnt.
// the four k-steps of one 32-column block of a [64][128] k-strided half-tile (256-B rows): f[ks] = rows ks*16 .. +15
template <int OFF>
__device__ __forceinline__ void frag4_tr_na(bf16x8_t (&f)[4], uint32_t addr) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x2_t lo, hi;
    switch (ks) {             // compile-time immediates
      case 0: lo = ds_read_tr_na<OFF>(addr); hi = ds_read_tr_na<OFF + 4 * 256>(addr); break;
      case 1: lo = ds_read_tr_na<OFF + 16 * 256>(addr); hi = ds_read_tr_na<OFF + 20 * 256>(addr); break;
      case 2: lo = ds_read_tr_na<OFF + 32 * 256>(addr); hi = ds_read_tr_na<OFF + 36 * 256>(addr); break;
      default: lo = ds_read_tr_na<OFF + 48 * 256>(addr); hi = ds_read_tr_na<OFF + 52 * 256>(addr); break;
    }
    f[ks] = tr_join(lo, hi);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---- activation epilogue, factored so that it can run in one go or be dribbled out under the next tile's K loop ------
// Per-lane state: the lane owns 8 consecutive output columns (n .. n+7) of every row it touches.
struct EpiLane {
  int n, cl; bool ncol_ok, wide;
  float bias8[8]; float csum[8];
};
// reload the bias strip (used by the dribbled epilogue, which cannot afford to keep it in registers across a K loop)
__device__ __forceinline__ void epi_load_bias(EpiLane& e, const GemmParams& p) {
#pragma unroll
  for (int q = 0; q < 8; ++q) e.bias8[q] = 0.f;
  if (p.bias && e.ncol_ok) {
    f32x4_t b = *(const f32x4_t*)(p.bias + e.n);
    e.bias8[0] = b[0]; e.bias8[1] = b[1]; e.bias8[2] = b[2]; e.bias8[3] = b[3];
    if (e.n + 4 < p.N) { f32x4_t c = *(const f32x4_t*)(p.bias + e.n + 4); e.bias8[4] = c[0]; e.bias8[5] = c[1]; e.bias8[6] = c[2]; e.bias8[7] = c[3]; }
  }
}
template <int WN>
__device__ __forceinline__ void epi_setup(EpiLane& e, const GemmParams& p, int lane, int col0) {
  constexpr int LPR = WN / 8;
  e.cl = (lane % LPR) * 8;
  e.n = col0 + e.cl;
  e.ncol_ok = e.n < p.N;                 // N % 4 == 0 is a host-checked precondition
  e.wide = p.wide_ok && (e.n + 8 <= p.N);
#pragma unroll
  for (int q = 0; q < 8; ++q) { e.bias8[q] = 0.f; e.csum[q] = 0.f; }
  if (p.bias && e.ncol_ok) {
    f32x4_t b = *(const f32x4_t*)(p.bias + e.n);
    e.bias8[0] = b[0]; e.bias8[1] = b[1]; e.bias8[2] = b[2]; e.bias8[3] = b[3];
    if (e.n + 4 < p.N) { f32x4_t c = *(const f32x4_t*)(p.bias + e.n + 4); e.bias8[4] = c[0]; e.bias8[5] = c[1]; e.bias8[6] = c[2]; e.bias8[7] = c[3]; }
  }
}
// W (8 or 4) consecutive columns starting at column offset `co` of the lane's strip; `src` = the strip's 8 values (fp32), row m
struct EpiStrip { u32x4_t w; };        // 8 bf16 of a second operand (saved derivative / residual) for one row strip
__device__ __forceinline__ EpiStrip epi_load_strip(const bf16_t* base, int ld, int m, const EpiLane& e, const GemmParams& p) {
  EpiStrip s; s.w = (u32x4_t){0u, 0u, 0u, 0u};
  const bf16_t* ptr = base + (size_t)m * ld + e.n;
  if (e.wide) s.w = *(const u32x4_t*)ptr;
  else {
    u32x2_t a = *(const u32x2_t*)ptr; s.w[0] = a[0]; s.w[1] = a[1];
    if (e.n + 4 < p.N) { u32x2_t b = *(const u32x2_t*)(ptr + 4); s.w[2] = b[0]; s.w[3] = b[1]; }
  }
  return s;
}
template <int W>
__device__ __forceinline__ void epi_cols(EpiLane& e, const GemmParams& p, const float (&src)[8], int co, int m,
                                         const EpiStrip& aux_s, const EpiStrip& res_s) {
  const int nn = e.n + co;
  float v[W];
  f32x2_t rs = {1.f, 0.f};
  if (p.ln_stat) rs = *(const f32x2_t*)(p.ln_stat + 2 * (size_t)m);
  if (p.ln_c) {                           // LayerNorm fold: rstd * acc + (bias - mean * rstd * c)
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = fmaf(src[co + k], rs[0], fmaf(rs[1], p.ln_c[nn + k], e.bias8[co + k]));
  } else {
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] = src[co + k] + e.bias8[co + k];
  }
  auto store_bf = [&](bf16_t* ptr, const float* x) {
    if (W == 8) {
      u32x4_t o; o[0] = pack2bf(x[0], x[1]); o[1] = pack2bf(x[2], x[3]); o[2] = pack2bf(x[4], x[5]); o[3] = pack2bf(x[6], x[7]);
      *(u32x4_t*)ptr = o;
    } else {
      u32x2_t o; o[0] = pack2bf(x[0], x[1]); o[1] = pack2bf(x[2], x[3]);
      *(u32x2_t*)ptr = o;
    }
  };
  if (p.act == 3) {                       // backward of an activation: multiply by the saved derivative
#pragma unroll
    for (int k = 0; k < W; ++k) { uint32_t w = aux_s.w[(co + k) >> 1]; v[k] *= ((co + k) & 1) ? bfhi(w) : bflo(w); }
  }
  if (p.act == 1 || p.act == 2) {         // GELU; the optional second output is GELU'(pre-activation) for backward
    float d[W];
#pragma unroll
    for (int k = 0; k < W; ++k) {
      if (p.act == 1) gelu_tab_scalar(v[k], v[k], d[k]); else gelu_tanh_both(v[k], v[k], d[k]);
    }
    if (p.C2) store_bf(p.C2 + (size_t)m * p.ldc2 + nn, d);
  } else if (p.C2) {
    store_bf(p.C2 + (size_t)m * p.ldc2 + nn, v);
  }
  if (p.drop_thresh) {
    const uint64_t ds = resolve_seed(p.drop_seed);
#pragma unroll
    for (int k = 0; k < W; ++k)
      v[k] = drop_keep(ds, (uint64_t)m * (uint64_t)p.N + (uint64_t)(nn + k), p.drop_thresh) ? v[k] * p.drop_scale : 0.f;
  }
  if (p.res) {
#pragma unroll
    for (int k = 0; k < W; ++k) { uint32_t w = res_s.w[(co + k) >> 1]; v[k] += ((co + k) & 1) ? bfhi(w) : bflo(w); }
  }
  if (p.colsum) {
#pragma unroll
    for (int k = 0; k < W; ++k) e.csum[co + k] += v[k];
  }
  if (p.ln_stat && !p.ln_c) {             // scale mode: the rows leave multiplied by rstd (the column sums above are over the unscaled values)
#pragma unroll
    for (int k = 0; k < W; ++k) v[k] *= rs[0];
  }
  if (p.out_f32) {
    float* crow = (float*)p.C + (size_t)m * p.ldc + nn;
#pragma unroll
    for (int q = 0; q < W / 4; ++q) *(f32x4_t*)(crow + 4 * q) = (f32x4_t){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  } else {
    store_bf((bf16_t*)p.C + (size_t)m * p.ldc + nn, v);
  }
}
// one output row (global row m): the lane's 8 values of it
// `prim` = the row's strip of the PRIMARY second operand (saved derivative when act == 3, else the residual): staged through
// LDS by LDS-DMA when `staged`, else (narrow leading dimensions) loaded here; a residual next to act == 3 is always loaded here
__device__ __forceinline__ void epi_row(EpiLane& e, const GemmParams& p, const float (&src)[8], int m, EpiStrip prim, bool staged) {
  if (m < p.M && e.ncol_ok) {
    EpiStrip res_s = prim;
    if (p.res && (p.act == 3 || !staged)) res_s = epi_load_strip(p.res, p.ldres, p.res_period ? (m % p.res_period) : m, e, p);
    if (p.act == 3 && !staged) prim = epi_load_strip(p.aux, p.ldaux, m, e, p);
    if (e.wide) epi_cols<8>(e, p, src, 0, m, prim, res_s);
    else {
      epi_cols<4>(e, p, src, 0, m, prim, res_s);
      if (e.n + 4 < p.N) epi_cols<4>(e, p, src, 4, m, prim, res_s);
    }
  }
}
template <int WN>
__device__ __forceinline__ void epi_flush_colsum(EpiLane& e, const GemmParams& p, int lane, int slot) {
  if (!p.colsum) return;
  constexpr int LPR = WN / 8;
  // lanes sharing (lane % LPR) own the same 8 columns: fold the row groups, one atomic per column
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) e.csum[q] += __shfl_xor(e.csum[q], o, 64);
  }
  if (lane < LPR && e.ncol_ok) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (e.n + q < p.N) {
        if (p.colsum_part) p.colsum_part[(size_t)slot * p.N + e.n + q] = e.csum[q];
        else unsafeAtomicAdd(&p.colsum[e.n + q], e.csum[q]);
      }
  }
}
// per-wave LDS of the activation epilogue: general path = fp32 patch [32][WN+4] + 2 second-operand buffers [32][WN] bf16,
// fast path = bias strip + 2 bf16 patches [32][WN*2+8 bytes] + 2 second-operand buffers [32][WN] bf16
template <int WN> constexpr int epi_wave_lds() {
  constexpr int general = 32 * (WN + 4) * 4 + 2 * 32 * WN * 2;
  constexpr int fast = WN * 4 + 2 * 32 * (WN * 2 + 8) + 2 * 32 * WN * 2;     // + 2 second-operand buffers
  return ((general > fast ? general : fast) + 15) & ~15;
}
// ---- general path: 32-row block of the wave tile -> wave-private fp32 LDS patch (4 consecutive columns = one ds_write_b128;
// rows 272 B apart, so the 8 lanes of a store group and the 2 rows of a load group sit on disjoint banks)
template <int TN> struct EpiBlk { f32x16_t t[TN]; };     // one 32-row block of the wave tile, passed by value (keeps the accumulators in registers)
template <int TM, int TN, int I>
__device__ __forceinline__ EpiBlk<TN> epi_take(const f32x16_t (&acc)[TM][TN]) {
  EpiBlk<TN> b;
#pragma unroll
  for (int j = 0; j < TN; ++j) b.t[j] = acc[I < TM ? I : 0][j];
  return b;
}
template <int TN, int WN>
__device__ __forceinline__ void epi_write_block(float* patch, const EpiBlk<TN> blk_, int lane) {
  const f32x16_t* blk = blk_.t;
  constexpr int LDP = WN + 4;
  float* dst = patch + (lane & 31) * LDP + 4 * (lane >> 5);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(f32x4_t*)(dst + j * 32 + 8 * q) = (f32x4_t){blk[j][4 * q], blk[j][4 * q + 1], blk[j][4 * q + 2], blk[j][4 * q + 3]};
}
template <int TM, int TN, int WN>
__device__ __forceinline__ void epi_write_block_i(float* patch, const f32x16_t (&acc)[TM][TN], int i, int lane) {
  switch (i) {          // accumulator registers need compile-time indices
    case 0: epi_write_block<TN, WN>(patch, epi_take<TM, TN, 0>(acc), lane); break;
    case 1: if (TM > 1) epi_write_block<TN, WN>(patch, epi_take<TM, TN, 1>(acc), lane); break;
    case 2: if (TM > 2) epi_write_block<TN, WN>(patch, epi_take<TM, TN, 2>(acc), lane); break;
    case 3: if (TM > 3) epi_write_block<TN, WN>(patch, epi_take<TM, TN, 3>(acc), lane); break;
    default: break;
  }
}

// 16-byte global stores of the epilogue outputs with a selectable L2 policy.  AVT_ST_AUX = 0: plain stores (the line stays in
// the XCD's L2: an M x 3072 activation written by one GEMM is consumed hundreds of microseconds later by another kernel, so all
// it does there is push the B operand out); 2 = nt (the line stays but is the first to go); 16 = sc1 (write-through, the line is
// dropped from L2 -- MI355X_MICROARCH.md, price list "stores of each flavour").  Measured on the whole step (256 clips, same box,
// profiles/r04_cache_policy.txt): plain 878.7 / 880.3 clips/s, sc1 883.6 / 882.7, nt 889.2 / 887.4 (+1.0 %; fc1 forward 3094 ->
// 2990 us with sc1) -> nt is the product's policy.  Rows are addressed relative to the wave tile's origin through a buffer descriptor.
constexpr int AVT_ST_AUX = 2;
struct TileStore {
  __amdgpu_buffer_rsrc_t r; bf16_t* base; int ld;
  __device__ __forceinline__ void init(bf16_t* origin, int ld_) {
    base = origin; ld = ld_;
    r = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, 0xFFFFFFF0u, 0x00020000);
  }
  __device__ __forceinline__ void st(int drow, int dcol, u32x4_t v) const {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)((drow * ld + dcol) * 2), 0, AVT_ST_AUX);
  }
};

// ---- fast path (bias / GELU (+ GELU') only, bf16 output): all arithmetic in the accumulator layout with packed fp32 ops,
// bf16 pairs through a [32][WN] bf16 patch (ds_write_b64 in, 16 B per lane out), one 16-byte global store per lane and row
template <int TN, int WN, int ACT, bool TAB = false>
__device__ __forceinline__ void epi_fast_block(const GemmParams& p, const EpiBlk<TN> blk_, char* patch_c, char* patch_d,
                                               const float* bias_l, int lane, int m0, int col0, const TileStore& sc, const TileStore& sd, int i32,
                                               const char* tab = nullptr) {
  const f32x16_t* blk = blk_.t;
  constexpr int LDB = WN * 2 + 8;                                  // patch row pitch (bytes): 16 store lanes -> 32 distinct banks
  const int ml = lane & 31, h = lane >> 5;
  if constexpr (TAB && ACT == 1) {
    // (the derivative patch is written whether or not the caller wants the second output: no per-element branch on it)
    u16x2_t mx = {0, 0};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      // all 16 look-ups of a 32-column slice are in flight together (one LDS round trip per slice, not per pair)
      uint32_t off[8], e[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = j * 32 + 8 * q + 4 * h;
        const f32x4_t b = *(const f32x4_t*)(bias_l + nl);
        const f32x2_t v0 = (f32x2_t){blk[j][4 * q], blk[j][4 * q + 1]} + (f32x2_t){b[0], b[1]};
        const f32x2_t v1 = (f32x2_t){blk[j][4 * q + 2], blk[j][4 * q + 3]} + (f32x2_t){b[2], b[3]};
        off[2 * q] = gelu_tab_offsets(pack2bf(v0[0], v0[1]), mx);
        off[2 * q + 1] = gelu_tab_offsets(pack2bf(v1[0], v1[1]), mx);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        e[2 * k] = *(const uint32_t*)(tab + (off[k] & 0xffffu));
        e[2 * k + 1] = *(const uint32_t*)(tab + (off[k] >> 16));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = j * 32 + 8 * q + 4 * h;
        const uint32_t h0 = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x05040100u), d0 = __builtin_amdgcn_perm(e[4 * q + 1], e[4 * q], 0x07060302u);
        const uint32_t h1 = __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x05040100u), d1 = __builtin_amdgcn_perm(e[4 * q + 3], e[4 * q + 2], 0x07060302u);
        *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){d0, d1};
        *(u32x2_t*)(patch_c + ml * LDB + nl * 2) = (u32x2_t){h0, h1};
      }
    }
    constexpr unsigned short HI = ((GELU_TAB_ELO + GELU_TAB_NEXP) << 7) - 1;
    if (__builtin_expect(__any((mx[0] > HI) | (mx[1] > HI)), 0)) {          // some value of this block lies above the table: patch those elements
#pragma unroll 1
      for (int j = 0; j < TN; ++j)
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
          const int nl = j * 32 + 8 * q + 4 * h;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = bf2f(f2bf(blk[j][4 * q + e] + bias_l[nl + e]));       // the bf16-rounded pre-activation, as in the look-up
            if (!(fabsf(x) < GELU_TAB_TOP)) {
              bf16_t hb, db; gelu_big(x, hb, db);
              *(bf16_t*)(patch_c + ml * LDB + (nl + e) * 2) = hb;
              if (p.C2) *(bf16_t*)(patch_d + ml * LDB + (nl + e) * 2) = db;
            }
          }
        }
    }
  } else
  {
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = j * 32 + 8 * q + 4 * h;
      const f32x4_t b = *(const f32x4_t*)(bias_l + nl);
      f32x2_t v0 = (f32x2_t){blk[j][4 * q], blk[j][4 * q + 1]} + (f32x2_t){b[0], b[1]};
      f32x2_t v1 = (f32x2_t){blk[j][4 * q + 2], blk[j][4 * q + 3]} + (f32x2_t){b[2], b[3]};
      if (ACT != 0) {
        f32x2_t d0, d1;
        if (ACT == 1) gelu_tab_both4(v0, v1, d0, d1);
        else {
          float y4[4], d4[4];
          gelu_tanh_both(v0[0], y4[0], d4[0]); gelu_tanh_both(v0[1], y4[1], d4[1]);
          gelu_tanh_both(v1[0], y4[2], d4[2]); gelu_tanh_both(v1[1], y4[3], d4[3]);
          v0 = (f32x2_t){y4[0], y4[1]}; v1 = (f32x2_t){y4[2], y4[3]}; d0 = (f32x2_t){d4[0], d4[1]}; d1 = (f32x2_t){d4[2], d4[3]};
        }
        if (p.C2) *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){pack2bf(d0[0], d0[1]), pack2bf(d1[0], d1[1])};
      }
      *(u32x2_t*)(patch_c + ml * LDB + nl * 2) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
    }
  }
  constexpr int LPR = WN / 8, RPI = 64 / LPR, IT = 32 / RPI;
  const int rl = lane / LPR, cl = (lane % LPR) * 8;
  const int n = col0 + cl;
  // with no activation the second output is the same tensor as the first (C2 = value before dropout, and the fast path has none)
  const char* patch_2 = (ACT != 0) ? patch_d : patch_c;
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int row = it * RPI + rl, m = m0 + row;
    const char* src = patch_c + row * LDB + cl * 2;
    u32x2_t lo = *(const u32x2_t*)src, hi = *(const u32x2_t*)(src + 8);
    u32x2_t lo2 = lo, hi2 = hi;
    if (p.C2) { const char* s2 = patch_2 + row * LDB + cl * 2; lo2 = *(const u32x2_t*)s2; hi2 = *(const u32x2_t*)(s2 + 8); }
    if (m < p.M && n < p.N) {
      sc.st(i32 + row, cl, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
      if (p.C2) sd.st(i32 + row, cl, (u32x4_t){lo2[0], lo2[1], hi2[0], hi2[1]});
    }
  }
}
template <int TM, int TN, int WN, int ACT, bool TAB = false>
__device__ __forceinline__ void epi_fast(const GemmParams& p, const f32x16_t (&acc)[TM][TN], char* wave_lds, int lane, int row0, int col0,
                                         const char* tab = nullptr) {
  constexpr int LDB = WN * 2 + 8;
  float* bias_l = (float*)wave_lds;
  char* patch_c = wave_lds + WN * 4;
  char* patch_d = patch_c + 32 * LDB;
  for (int c_ = lane; c_ < WN; c_ += 64) bias_l[c_] = (p.bias && col0 + c_ < p.N) ? p.bias[col0 + c_] : 0.f;
  TileStore sc, sd;
  sc.init((bf16_t*)p.C + (size_t)row0 * p.ldc + col0, p.ldc);
  sd.init(p.C2 ? p.C2 + (size_t)row0 * p.ldc2 + col0 : (bf16_t*)p.C, p.ldc2);
  if (ACT == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      EpiBlk<TN> b;
#pragma unroll
      for (int j = 0; j < TN; ++j) b.t[j] = acc[i][j];
      epi_fast_block<TN, WN, ACT>(p, b, patch_c, patch_d, bias_l, lane, row0 + i * 32, col0, sc, sd, i * 32);
    }
  } else {
    // the GELU arithmetic of one block is ~1.5 k instructions: keep ONE copy of it (instruction cache) and move the block
    // into place instead (32 register copies per block)
#pragma unroll 1
    for (int i = 0; i < TM; ++i) {
      EpiBlk<TN> b;
      switch (i) {
        case 0: b = epi_take<TM, TN, 0>(acc); break;
        case 1: b = epi_take<TM, TN, 1>(acc); break;
        case 2: b = epi_take<TM, TN, 2>(acc); break;
        default: b = epi_take<TM, TN, 3>(acc); break;
      }
      epi_fast_block<TN, WN, ACT, TAB>(p, b, patch_c, patch_d, bias_l, lane, row0 + i * 32, col0, sc, sd, i * 32, tab);
    }
  }
}

// ---- extended fast path: the same register-layout arithmetic plus everything that needs a second operand or an index:
//   v = acc + bias;  act 3: v *= aux;  act 1|2: GELU (+ GELU' -> C2);  act 0 with C2: C2 = v;  dropout;  v += res;
//   column sums;  C = bf16(v)
// The second operand (aux when act == 3, else res) of block i is brought in by LDS-DMA two blocks ahead.  LDS-DMA lands
// lane L's 16 bytes at buffer + 16 L, so the layout is chosen through the SOURCE address: position (row r, chunk pc) holds
// chunk pc ^ ((r >> 1) & (LPR-1)) of row r, which makes the 8-byte reads of the accumulator layout (32 lanes = 32 rows, same
// column) at most 2-way bank conflicted.  Column sums (bias gradients) are taken over the bf16-rounded outputs on their way
// out (row-strip layout: 8 running sums per lane, folded across the lanes that share columns once per tile).
// LNM (LayerNorm-fold mode, compile time: the 8-phase kernels sit at their register limit): 0 none | 1 fold (ACT 0 | 1) | 2 scale (ACT 3) | 3 row statistics out (ACT 0)
template <int TM, int TN, int WN, int ACT, int LNM = 0>
__device__ __forceinline__ void epi_fast_ext(const GemmParams& p, const f32x16_t (&acc)[TM][TN], char* wave_lds, int lane, int row0, int col0) {
  static_assert(LNM == 0 || (LNM == 1 && (ACT == 0 || ACT == 1)) || (LNM == 2 && ACT == 3) || (LNM == 3 && ACT == 0), "epi_fast_ext: LayerNorm-fold mode");
  constexpr int LDB = WN * 2 + 8;
  constexpr int LPR = WN / 8, RPI = 64 / LPR, IT = 32 / RPI;
  constexpr int OPB = 32 * WN * 2;
  float* bias_l = (float*)wave_lds;
  char* patch_c = wave_lds + WN * 4;
  char* patch_d = patch_c + 32 * LDB;
  char* opbuf = patch_d + 32 * LDB;
  constexpr bool fold = LNM == 1;                              // LayerNorm fold (host-checked: no second operand then, so its buffer holds the c strip)
  constexpr bool scale = LNM == 2;                             // rows leave multiplied by rstd, column sums over the unscaled values
  constexpr bool stats = LNM == 3;
  float* c_l = (float*)opbuf;
  for (int c_ = lane; c_ < WN; c_ += 64) {
    bias_l[c_] = (p.bias && col0 + c_ < p.N) ? p.bias[col0 + c_] : 0.f;
    if (fold) c_l[c_] = (col0 + c_ < p.N) ? p.ln_c[col0 + c_] : 0.f;
  }
  const int ml = lane & 31, h = lane >> 5;
  const int rl = lane / LPR, pc = lane % LPR, cl = pc * 8;
  // the lane's row statistics for its row of every 32-row block (accumulator layout: lane = row); requested before the second operand's DMA so
  // that the counted waits below (operations YOUNGER than DMA(i)) stay as they are
  f32x2_t rst[(fold || scale) ? TM : 1];
  if constexpr (fold || scale) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      rst[i] = (f32x2_t){1.f, 0.f};
      const int m = row0 + i * 32 + ml;
      if (m < p.M) rst[i] = *(const f32x2_t*)(p.ln_stat + 2 * (size_t)m);
    }
  }
  const bf16_t* prim_ptr = (ACT == 3) ? p.aux : p.res;
  const int prim_ld = (ACT == 3) ? p.ldaux : p.ldres;
  const int prim_period = (ACT == 3) ? 0 : p.res_period;
  const bool has_prim = prim_ptr != nullptr;
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)prim_ptr, 0, 0xFFFFFFF0u, 0x00020000);
  auto dma_block = [&](int i) __attribute__((always_inline)) {
#pragma unroll
    for (int itr = 0; itr < IT; ++itr) {
      const int r = itr * RPI + rl;
      const int m = row0 + i * 32 + r;
      const int n = col0 + ((pc ^ ((r >> 1) & (LPR - 1))) * 8);
      const int mr = prim_period ? (m % prim_period) : m;
      uint32_t off = (uint32_t)(((size_t)mr * (size_t)prim_ld + (size_t)n) * 2);
      if (m >= p.M || n >= p.N) off = 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, AVT_LDS_PTR(opbuf + (i & 1) * OPB + itr * 1024), 16, off, 0, 0, 0);
    }
  };
  if (has_prim) { dma_block(0); if (TM > 1) dma_block(1); }
  TileStore sc, sd;
  sc.init((bf16_t*)p.C + (size_t)row0 * p.ldc + col0, p.ldc);
  sd.init(p.C2 ? p.C2 + (size_t)row0 * p.ldc2 + col0 : (bf16_t*)p.C, p.ldc2);
  float cs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cs[k] = 0.f;
  constexpr int UNR = (ACT == 0 || ACT == 3) ? 4 : 1;
#pragma unroll UNR
  for (int i = 0; i < TM; ++i) {
    EpiBlk<TN> b;
    f32x2_t rs = {1.f, 0.f};
    constexpr bool RS = fold || scale;
    switch (i) {
      case 0: b = epi_take<TM, TN, 0>(acc); if (RS) rs = rst[0]; break;
      case 1: b = epi_take<TM, TN, 1>(acc); if (RS) rs = rst[RS && TM > 1 ? 1 : 0]; break;
      case 2: b = epi_take<TM, TN, 2>(acc); if (RS) rs = rst[RS && TM > 2 ? 2 : 0]; break;
      default: b = epi_take<TM, TN, 3>(acc); if (RS) rs = rst[RS && TM > 3 ? 3 : 0]; break;
    }
    if (has_prim) {          // see the general path for the counts
      switch (i) {
        case 0: if (TM > 1) wait_vmcnt<IT>(); else wait_vmcnt<0>(); break;
        case 1: if (TM > 2) wait_vmcnt<2 * IT>(); else wait_vmcnt<IT>(); break;
        case 2: if (TM > 3) wait_vmcnt<3 * IT>(); else wait_vmcnt<2 * IT>(); break;
        default: wait_vmcnt<2 * IT>(); break;
      }
    }
    const int m = row0 + i * 32 + ml;
    u32x2_t opv[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        opv[j][q] = (u32x2_t){0u, 0u};
        if (has_prim) opv[j][q] = *(const u32x2_t*)(opbuf + (i & 1) * OPB + ml * (WN * 2) + (((j * 4 + q) ^ ((ml >> 1) & (LPR - 1))) * 16) + h * 8);
      }
    if (has_prim && i + 2 < TM) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer's previous contents are in registers
      dma_block(i + 2);
    }
    const f32x2_t rr = (f32x2_t){rs[0], rs[0]}, tt = (f32x2_t){rs[1], rs[1]};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x2_t s1 = {0.f, 0.f}, s2 = {0.f, 0.f};               // the row's partial (sum, sum of squares) over this 32-column slot (stat_part)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = j * 32 + 8 * q + 4 * h;
        const f32x4_t bb = *(const f32x4_t*)(bias_l + nl);
        f32x2_t v0, v1;
        if (fold) {
          const f32x4_t cc = *(const f32x4_t*)(c_l + nl);
          v0 = (f32x2_t){b.t[j][4 * q], b.t[j][4 * q + 1]} * rr + ((f32x2_t){cc[0], cc[1]} * tt + (f32x2_t){bb[0], bb[1]});
          v1 = (f32x2_t){b.t[j][4 * q + 2], b.t[j][4 * q + 3]} * rr + ((f32x2_t){cc[2], cc[3]} * tt + (f32x2_t){bb[2], bb[3]});
        } else {
          v0 = (f32x2_t){b.t[j][4 * q], b.t[j][4 * q + 1]} + (f32x2_t){bb[0], bb[1]};
          v1 = (f32x2_t){b.t[j][4 * q + 2], b.t[j][4 * q + 3]} + (f32x2_t){bb[2], bb[3]};
        }
        const f32x2_t o0 = (f32x2_t){bflo(opv[j][q][0]), bfhi(opv[j][q][0])}, o1 = (f32x2_t){bflo(opv[j][q][1]), bfhi(opv[j][q][1])};
        if (ACT == 3) { v0 *= o0; v1 *= o1; if (scale) { v0 *= rr; v1 *= rr; } }
        if (ACT == 1 || ACT == 2) {
          f32x2_t d0, d1;
          if (ACT == 1) gelu_tab_both4(v0, v1, d0, d1);
          else {
            float y4[4], d4[4];
            gelu_tanh_both(v0[0], y4[0], d4[0]); gelu_tanh_both(v0[1], y4[1], d4[1]);
            gelu_tanh_both(v1[0], y4[2], d4[2]); gelu_tanh_both(v1[1], y4[3], d4[3]);
            v0 = (f32x2_t){y4[0], y4[1]}; v1 = (f32x2_t){y4[2], y4[3]}; d0 = (f32x2_t){d4[0], d4[1]}; d1 = (f32x2_t){d4[2], d4[3]};
          }
          if (p.C2) *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){pack2bf(d0[0], d0[1]), pack2bf(d1[0], d1[1])};
        } else if (p.C2) {
          *(u32x2_t*)(patch_d + ml * LDB + nl * 2) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
        }
        if (p.drop_thresh) {
          const uint64_t idx = (uint64_t)m * (uint64_t)p.N + (uint64_t)(col0 + nl), ds = resolve_seed(p.drop_seed);
          v0[0] = drop_keep(ds, idx, p.drop_thresh) ? v0[0] * p.drop_scale : 0.f;
          v0[1] = drop_keep(ds, idx + 1, p.drop_thresh) ? v0[1] * p.drop_scale : 0.f;
          v1[0] = drop_keep(ds, idx + 2, p.drop_thresh) ? v1[0] * p.drop_scale : 0.f;
          v1[1] = drop_keep(ds, idx + 3, p.drop_thresh) ? v1[1] * p.drop_scale : 0.f;
        }
        if (ACT != 3 && has_prim) { v0 += o0; v1 += o1; }
        if (stats) { s1 += v0; s1 += v1; s2 += v0 * v0; s2 += v1 * v1; }
        *(u32x2_t*)(patch_c + ml * LDB + nl * 2) = (u32x2_t){pack2bf(v0[0], v0[1]), pack2bf(v1[0], v1[1])};
      }
      if (stats) {
        // the two half-waves hold the two halves of the slot's columns of the same row
        float a1 = s1[0] + s1[1], a2 = s2[0] + s2[1];
        a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
        if (h == 0 && m < p.M && col0 + j * 32 < p.N)
          *(f32x2_t*)(p.stat_part + ((size_t)((col0 + j * 32) >> 6) * (size_t)p.M + (size_t)m) * 4 + (((col0 + j * 32) >> 5) & 1) * 2) = (f32x2_t){a1, a2};
      }
    }
    const int n = col0 + cl;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int row = it * RPI + rl, mm = row0 + i * 32 + row;
      const char* src = patch_c + row * LDB + cl * 2;
      u32x2_t lo = *(const u32x2_t*)src, hi = *(const u32x2_t*)(src + 8);
      u32x2_t lo2 = lo, hi2 = hi;
      if (p.C2) { const char* s2_ = patch_d + row * LDB + cl * 2; lo2 = *(const u32x2_t*)s2_; hi2 = *(const u32x2_t*)(s2_ + 8); }
      float w = 1.f;
      if (scale) w = __shfl(rs[1], row, 64);                  // 1 / rstd of the strip's row: the column sums are over the unscaled values
      if (mm < p.M && n < p.N) {
        sc.st(i * 32 + row, cl, (u32x4_t){lo[0], lo[1], hi[0], hi[1]});
        if (p.C2) sd.st(i * 32 + row, cl, (u32x4_t){lo2[0], lo2[1], hi2[0], hi2[1]});
        if (p.colsum) {
          cs[0] += w * bflo(lo[0]); cs[1] += w * bfhi(lo[0]); cs[2] += w * bflo(lo[1]); cs[3] += w * bfhi(lo[1]);
          cs[4] += w * bflo(hi[0]); cs[5] += w * bfhi(hi[0]); cs[6] += w * bflo(hi[1]); cs[7] += w * bfhi(hi[1]);
        }
      }
    }
  }
  if (p.colsum) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) cs[q] += __shfl_xor(cs[q], o, 64);
    }
    if (lane < LPR && col0 + cl < p.N) {
      if (p.colsum_part) {
        float* dst = p.colsum_part + (size_t)(row0 / (TM * 32)) * p.N + col0 + cl;
        *(f32x4_t*)dst = (f32x4_t){cs[0], cs[1], cs[2], cs[3]};
        *(f32x4_t*)(dst + 4) = (f32x4_t){cs[4], cs[5], cs[6], cs[7]};
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) unsafeAtomicAdd(&p.colsum[col0 + cl + q], cs[q]);
      }
    }
  }
}

// split-K slab stores (written once, read once by splitk_reduce_kernel): plain -- nontemporal measured +0.15 % (noise), profiles/r05e_w4_cache_policy.txt
#define AVT_SLAB_ST(ptr, val) (*(f32x4_t*)(ptr) = (val))
// LN: compile the LayerNorm-fold variants of the epilogue (only the kernels with both operands k-major are ever asked for them)
// DIRECT (EPI 2): 1 = the workgroup holds the whole reduction (splitk == 1) and adds its tile into C itself, 2 = ... writes it (GemmParams::c_assign), 0 = slabs.  Only the 4-wave weight-gradient
// kernel has the direct form, as a separate instantiation (both in one body spilled 16 of its registers; in the generic 128 x 128 kernel the 64 values in
// flight would cost the second workgroup per CU): the head's 2048 x 8192 weights are what it is for.
template <int TM, int TN, int WM, int WN, int EPI, int PR = 0, bool TAB = false, bool LN = false, int DIRECT = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16_t (&acc)[TM][TN], char* lds, int wave, int lane,
                                              int row0, int col0, const char* tab = nullptr) {
  // row0/col0: global coordinates of this wave's tile origin
  static_assert(WM == TM * 32 && WN == TN * 32 && TM <= 4, "wave tile geometry");
  if (EPI == 2) {
    if constexpr (DIRECT >= 1) {
      // (round 6) the whole reduction sits in this workgroup: C += acc right here -- every C element has exactly one owner, plain read-modify-write,
      // the same sum bit for bit as a one-slab reduce -- instead of a slab round trip (tile written, re-read, C read and written) and a second launch.
      // Each store instruction covers two rows x 32 consecutive columns: two full 128-byte lines.  (The head's 2048 x 8192 weights at <= 2560 rows.)
      // Branch-free, batched: all the loads of a 32-row block row (TN x 16 per lane) are issued before the first add, and elements past M / N are
      // kept out by the buffer descriptor's bounds check (an out-of-range offset: loads return 0, stores are dropped) instead of a predicate.  The
      // first version -- a plain `if (in range) C[..] += acc` loop -- compiled to s_waitcnt vmcnt(0) in front of every one of the 256 stores per wave
      // (hipcc waits for everything outstanding at a branch that hides the count) and cost +95 us per launch (profiles/r06e_direct_accumulate_ab.txt).
      const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (uint32_t)((size_t)p.M * (size_t)p.ldc * 4), 0x00020000);   // (host-checked: < 4 GiB)
      const int nl = lane & 31, hq = 4 * (lane >> 5);
      const uint32_t ld4 = (uint32_t)p.ldc * 4u;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float old[TN][16];
        uint32_t off[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = col0 + j * 32 + nl;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int m0 = row0 + i * 32 + 8 * q + hq;
            off[j][q] = n < p.N ? (uint32_t)m0 * ld4 + (uint32_t)n * 4u : 0xFFFFFFF0u;      // (rows past M end up past the descriptor's range by themselves)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              old[j][4 * q + k] = DIRECT == 2 ? 0.f      // (assign: C is known to hold zeros -- 0 + acc = acc bit for bit, accumulators never hold -0)
                                              : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off[j][q] == 0xFFFFFFF0u ? off[j][q] : off[j][q] + (uint32_t)k * ld4, 0, 0));
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, old[j][4 * q + k] + acc[i][j][4 * q + k]), rc,
                                                    off[j][q] == 0xFFFFFFF0u ? off[j][q] : off[j][q] + (uint32_t)k * ld4, 0, 0);
      }
      return;
    }
    // deterministic weight-gradient epilogue: this block's partial tile goes to its own slab of the caller's workspace in
    // accumulator order (full 1-KB wave stores); splitk_reduce_kernel adds the slabs in split order into C
    const int NWV = (int)(blockDim.x >> 6);
    const int slab = accum_slab(p);                                                 // = split * tiles + tile
    float* dst = p.ws + (size_t)slab * (size_t)(NWV * TM * TN * 1024) + (size_t)wave * (TM * TN * 1024) + lane * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          AVT_SLAB_ST(dst + ((i * TN + j) * 4 + q) * 256, ((f32x4_t){acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]}));
    return;
  } else if (EPI == 1) {
    // weight-gradient epilogue: fp32 accumulate into C (atomics; C is pre-zeroed or holds the running sum)
    float* C = (float*)p.C;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int n = col0 + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < p.M && n < p.N) unsafeAtomicAdd(&C[(size_t)m * p.ldc + n], acc[i][j][r]);
        }
      }
    return;
  } else {
    // The LDS pipe executes one wave's operations in order, so a wave-private patch needs no wait between writing it and
    // reading it back, and a patch can be rewritten as soon as the reads of its previous contents have been ISSUED.
    constexpr int LDP = WN + 4;
    char* wave_lds = lds + wave * epi_wave_lds<WN>();
    const bool fast_ok = !p.out_f32 && p.wide_ok && (p.N % 8 == 0) && !(p.act == 3 && p.res);
    if (fast_ok) {
      const bool extra = p.res || p.act == 3 || p.colsum || p.drop_thresh || p.ln_stat || p.stat_part;
      if (!extra) {
        if (p.act == 0) epi_fast<TM, TN, WN, 0>(p, acc, wave_lds, lane, row0, col0);
        else if (p.act == 1) epi_fast<TM, TN, WN, 1, TAB>(p, acc, wave_lds, lane, row0, col0, tab);
        else epi_fast<TM, TN, WN, 2>(p, acc, wave_lds, lane, row0, col0);
      } else if (p.act == 3) {
        if (LN && p.ln_stat) epi_fast_ext<TM, TN, WN, 3, LN ? 2 : 0>(p, acc, wave_lds, lane, row0, col0);
        else epi_fast_ext<TM, TN, WN, 3>(p, acc, wave_lds, lane, row0, col0);
      } else if (p.act == 0) {
        if (LN && p.ln_c) epi_fast_ext<TM, TN, WN, 0, LN ? 1 : 0>(p, acc, wave_lds, lane, row0, col0);
        else if (LN && p.stat_part) epi_fast_ext<TM, TN, WN, 0, LN ? 3 : 0>(p, acc, wave_lds, lane, row0, col0);
        else epi_fast_ext<TM, TN, WN, 0>(p, acc, wave_lds, lane, row0, col0);
      } else if (p.act == 1) {
        if (LN && p.ln_c) epi_fast_ext<TM, TN, WN, 1, LN ? 1 : 0>(p, acc, wave_lds, lane, row0, col0);
        else epi_fast_ext<TM, TN, WN, 1>(p, acc, wave_lds, lane, row0, col0);
      } else epi_fast_ext<TM, TN, WN, 2>(p, acc, wave_lds, lane, row0, col0);
      return;
    }
    // general path, per 32-row block i: wait for the second operand of block i (LDS-DMA issued two blocks earlier: global ->
    // LDS, no registers, lane l's 16 bytes land at buffer + 16 l, exactly its row strip), request it and the rows of patch i
    // from LDS, start the DMA of block i+2 into the buffer just read, queue the patch writes of block i+1, then do the
    // arithmetic and the global stores of block i.
    constexpr int LPR = WN / 8;          // lanes per row
    constexpr int RPI = 64 / LPR;        // rows per iteration = rows per DMA instruction
    constexpr int IT = 32 / RPI;
    float* patch = (float*)wave_lds;
    char* opbuf = wave_lds + 32 * LDP * 4;
    constexpr int OPB = 32 * WN * 2;     // one buffer: 32 rows of the operand
    EpiLane e;
    epi_setup<WN>(e, p, lane, col0);
    const bf16_t* prim_ptr = (p.act == 3) ? p.aux : p.res;
    const int prim_ld = (p.act == 3) ? p.ldaux : p.ldres;
    const int prim_period = (p.act == 3) ? 0 : p.res_period;
    const bool staged = prim_ptr != nullptr && p.wide_ok && (p.N % 8 == 0);
    __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)prim_ptr, 0, 0xFFFFFFF0u, 0x00020000);
    const int rl = lane / LPR;
    auto dma_block = [&](int i) __attribute__((always_inline)) {
#pragma unroll
      for (int itr = 0; itr < IT; ++itr) {
        const int m = row0 + i * 32 + itr * RPI + rl;
        const int mr = prim_period ? (m % prim_period) : m;
        uint32_t off = (uint32_t)(((size_t)mr * (size_t)prim_ld + (size_t)e.n) * 2);
        if (m >= p.M || !e.ncol_ok) off = 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rp, AVT_LDS_PTR(opbuf + (i & 1) * OPB + itr * 1024), 16, off, 0, 0, 0);
      }
    };
    if (staged) { dma_block(0); if (TM > 1) dma_block(1); }
    epi_write_block<TN, WN>(patch, epi_take<TM, TN, 0>(acc), lane);
#pragma unroll 1
    for (int i = 0; i < TM; ++i) {
      // outstanding VMEM operations issued after DMA(i), counting one store per row strip (more stores only make the wait
      // stricter than needed): i = 0: DMA(1);  i = 1: DMA(2), stores(0);  i >= 2: stores(i-2), DMA(i+1), stores(i-1)
      if (staged) {
        switch (i) {
          case 0: if (TM > 1) wait_vmcnt<IT>(); else wait_vmcnt<0>(); break;
          case 1: if (TM > 2) wait_vmcnt<2 * IT>(); else wait_vmcnt<IT>(); break;
          case 2: if (TM > 3) wait_vmcnt<3 * IT>(); else wait_vmcnt<2 * IT>(); break;
          default: wait_vmcnt<2 * IT>(); break;
        }
      }
      float rows[IT][8];
      EpiStrip prim[IT];
#pragma unroll
      for (int itr = 0; itr < IT; ++itr) {
        const float* src = patch + (itr * RPI + rl) * LDP + e.cl;
        f32x4_t lo = *(const f32x4_t*)src, hi = *(const f32x4_t*)(src + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { rows[itr][k] = lo[k]; rows[itr][4 + k] = hi[k]; }
        prim[itr].w = (u32x4_t){0u, 0u, 0u, 0u};
        if (staged) prim[itr].w = *(const u32x4_t*)(opbuf + (i & 1) * OPB + itr * 1024 + lane * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (staged && i + 2 < TM) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer's previous contents are in registers
        dma_block(i + 2);
      }
      if (i + 1 < TM) epi_write_block_i<TM, TN, WN>(patch, acc, i + 1, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int itr = 0; itr < IT; ++itr)
        epi_row(e, p, rows[itr], row0 + i * 32 + itr * RPI + rl, prim[itr], staged);
    }
    epi_flush_colsum<WN>(e, p, lane, row0 / WM);
  }
}

}  // namespace
